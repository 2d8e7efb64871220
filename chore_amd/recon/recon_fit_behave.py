"""SMPL-H + object fitting loops (BEHAVE protocol) on the GPU.

Counterpart of /root/reference/recon/recon_fit_behave.py: `optimize_smpl` (:224-291), `forward_smpl`
(:293-337), `optimize_smpl_object` (:90-163), `forward_step` (:165-222), `get_loss_weights` (:339-358).
The schedules, optimiser re-creation points, the weight formula w*L/(1+decay) and the reference's
quirks are kept on purpose because they change the fitted poses:
  * zero_grad() once per OUTER iteration, then `steps_per_iter` x {backward; step} -> gradients accumulate
    over the inner steps;
  * a new Adam at every phase switch; in `optimize_smpl_object` the SMPL parameters are never stepped;
  * `forward_step` queries the object points twice per step (once directly, once in compute_obj_loss).
The 'sil' phase runs if the caller provides data_dict['silhouette'] (SilLossROI); the 'collide' term of the joint
phase (recon_fit_behave.py:213-216) runs if the fitter was given the object template mesh (scan_verts / scan_faces).
Per-step host synchronisation of the reference (tqdm strings, .item()) is gone: the early-stop test
reads the loss once per OUTER iteration.
"""
import torch
import torch.nn.functional as F

from ..lib_smpl.const import SMPL_POSE_PRAMS_NUM
from .graph_step import EagerStep, GraphedStep
from .recon_fit_base import ReconFitterBase


class ReconFitterBehave(ReconFitterBase):
    use_graphs = False    # True: every inner step is a hipGraph replay (graph_step.py); same update rule
    adam_capturable = False   # eager steps with Adam's scalars evaluated on the device (what the graph does)

    def _stepper(self, *a, **k):
        if self.use_graphs:
            return GraphedStep(*a, **k)
        return EagerStep(*a, capturable=self.adam_capturable, **k)

    @staticmethod
    def release_graphs(split, model):
        """forget what holds autograd graphs of earlier steps (see GraphedStep)"""
        def f():
            split.betas = split.pose = None
            split.forget()
            model.preds = None
            model.points = None            # CHORE.query keeps its last inputs (reference attribute): a graph too
            model.intermediate_preds_list = []
        return f

    def get_loss_weights(self):
        w = {"beta": 1.0, "pose": 1e-5, "hand": 1e-5, "j2d": 0.3 ** 2, "object": 30.0 ** 2, "part": 0.05 ** 2,
             "contact": 30.0 ** 2, "scale": 10.0 ** 2, "df_h": 30.0 ** 2, "smplz": 30 ** 2, "mask": 0.003 ** 2,
             "ocent": 15 ** 2, "collide": 3 ** 2, "pinit": 5 ** 2, "rot": 10.0 ** 2, "trans": 10.0 ** 2}
        return {k: (lambda cst, it, c=c: c * cst / (1 + it)) for k, c in w.items()}

    # ---- SMPL ---------------------------------------------------------------------------------------
    def forward_smpl(self, smpl, data_dict, phase):
        loss_dict = {}
        model = data_dict["net"]
        smpl.forget()   # the LBS memo lives for one step (its autograd graph is consumed by this step's backward)
        smpl_verts = smpl()[0]
        _, parts_pred, _ = self.compute_df_h_loss(data_dict, loss_dict, model, smpl_verts)
        self.compute_prior_loss(loss_dict, smpl, nobeta=True)
        loss_dict["part"] = F.cross_entropy(parts_pred, data_dict["part_labels"], reduction="none").sum(-1).mean()
        J, _, _ = smpl.get_landmarks()
        self.smplz_loss(J, loss_dict)
        loss_dict["pinit"] = torch.mean(torch.sum((smpl.pose[:, 3:SMPL_POSE_PRAMS_NUM] - data_dict["pose_init"]) ** 2, -1))
        if phase == "kpts":
            self.compute_kpts_loss(data_dict, loss_dict, smpl)
        return loss_dict

    def optimize_smpl(self, smpl, data_dict, iter_for_betas=10, iter_for_pose=10, iter_for_kpts=5, steps_per_iter=10,
                      max_iter=150):
        split = self.split_smpl(smpl)
        height_init = self.get_smpl_height(smpl)
        wd = self.get_loss_weights()
        prev = torch.tensor(300.0, device=self.device)

        def loss_of(phase):
            return lambda decay: self.sum_dict(self.forward_smpl(split, data_dict, phase), wd, decay)

        phase = "global"
        rel = self.release_graphs(split, data_dict["net"])
        st = self._stepper([split.top_betas, split.trans], 0.02, loss_of(phase), 0.001, prev, release=rel)
        for it in range(iter_for_betas + iter_for_kpts + iter_for_pose + max_iter):
            if it == iter_for_betas:
                phase = "smpl all pose"   # new Adam
                st = self._stepper([split.trans, split.global_pose, split.body_pose, split.top_betas, split.other_betas],
                                   0.006, loss_of(phase), 0.001, prev, release=rel)
            elif it == iter_for_betas + iter_for_pose:
                phase = "kpts"            # same Adam, the loss gains the keypoint term
                st = self._stepper(st.params, 0.006, loss_of(phase), 0.001, prev, opt=st.opt, release=rel)
            st.begin_outer(1 if phase != "kpts" else it / 3)
            for _ in range(steps_per_iter):
                st.step()
            if it > 0.25 * max_iter + iter_for_betas + iter_for_pose and st.stopped():
                break
        rel()   # graph replays change the parameters without touching their version counters: drop memoised results
        scale = self.get_smpl_height(split) / height_init
        return self.copy_smpl_params(split, smpl), scale

    # ---- object + joint -----------------------------------------------------------------------------
    def forward_step(self, model, smpl, data_dict, obj_R, obj_t, obj_s, phase, noise=None):
        smpl.forget()
        smpl_verts = smpl()[0]
        loss_dict = {}
        R = self.decopose_axis(obj_R, noise=noise)
        object = self.transform_obj_verts(data_dict["objects"], R, obj_t, obj_s)
        model.query(object, **data_dict["query_dict"])
        df_pred, _, part_o, centers_o = model.get_preds()
        obj_center_pred = data_dict["smpl_center"] + torch.mean(centers_o[:, 3:, :], -1)
        if phase == "sil":
            obj_losses = data_dict["silhouette"](R, obj_t, obj_s)[0]
            loss_dict["mask"] = obj_losses["mask"]
            loss_dict["scale"] = torch.mean((obj_s - self.obj_scale) ** 2)
            loss_dict["trans"] = torch.mean((obj_t - data_dict["trans_init"]) ** 2)
            return loss_dict
        self.compute_obj_loss(data_dict, loss_dict, model, obj_s, object)
        loss_dict["ocent"] = F.mse_loss(torch.mean(object, 1), obj_center_pred, reduction="none").sum(-1).mean()
        if phase == "joint":
            df_obj_h = df_pred[:, 0, :]
            model.query(smpl_verts, **data_dict["query_dict"])
            df_hum_o = model.get_preds()[0][:, 1, :]
            self.compute_contact_loss(df_hum_o, df_obj_h, object, smpl_verts, loss_dict, part_o=part_o)
            if self.scan_faces is not None:
                loss_dict["collide"] = self.compute_collision_loss(smpl_verts, smpl.faces, R, obj_t, obj_s)
        return loss_dict

    def optimize_smpl_object(self, model, data_dict, obj_iter=20, joint_iter=10, steps_per_iter=10, sil_iter=50,
                             max_iter=100):
        smpl = data_dict["smpl"]
        split = self.split_smpl(smpl)
        data_dict["smpl"] = split
        obj_R, obj_t, obj_s = data_dict["obj_R"], data_dict["obj_t"], data_dict["obj_s"]
        wd = self.get_loss_weights()
        if "silhouette" not in data_dict:
            sil_iter = 0
        data_dict["smpl_center"] = self.compute_smpl_center_pred(data_dict, model, smpl)
        prev = torch.tensor(300.0, device=self.device)
        n_outer = joint_iter + obj_iter + max_iter + sil_iter
        # the SO(3) perturbation of every step (recon_fit_base.py:384) is drawn from the CPU generator like the
        # reference does, but for all steps at once (one call yields the same stream as one call per step) so that
        # a step only reads noise[k] on the device
        B = obj_R.shape[0]
        noise = torch.rand(n_outer * steps_per_iter, B, 3, 3).to(self.device)
        k = torch.zeros(1, dtype=torch.long, device=self.device)

        def loss_of(phase):
            def f(decay):
                nz = noise.index_select(0, k).squeeze(0)
                k.add_(1)
                return self.sum_dict(self.forward_step(model, split, data_dict, obj_R, obj_t, obj_s, phase, noise=nz), wd, decay)
            return f

        phase = "object only"
        rel = self.release_graphs(split, model)
        st = self._stepper([obj_t, obj_R, obj_s], 0.006, loss_of(phase), 0.0001, prev, state=[k], release=rel)
        for it in range(n_outer):
            if it == obj_iter and sil_iter > 0:
                phase = "sil"
                st = self._stepper([obj_R, obj_s, obj_t], 0.006, loss_of(phase), 0.0001, prev, state=[k], release=rel)
                data_dict["rot_init"] = self.decopose_axis(obj_R).detach().clone()
                data_dict["trans_init"] = obj_t.detach().clone()
            if it == obj_iter + sil_iter:
                phase = "joint"
                st = self._stepper([obj_t, obj_s], 0.002, loss_of(phase), 0.0001, prev, state=[k], release=rel)
            decay = 1 if phase == "object only" else it
            if phase == "sil":
                decay = it - obj_iter + 1
            elif phase == "joint":
                decay = (it - obj_iter + 1) / 5
            st.begin_outer(decay)
            for _ in range(steps_per_iter):
                st.step()
            if phase == "joint" and it > 0.25 * max_iter and st.stopped():
                break
        rel()
        return smpl, data_dict["obj_R"], data_dict["obj_t"]
