"""GPU: SO(3) projection parity (against torch.svd on CPU = what the reference calls) and the fit
iterations of recon_fit_behave on synthetic SMPL-H / object data."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def ref_project_so3(mat):
    """the reference's formula, verbatim semantics (recon/recon_fit_base.py:168-188), on CPU"""
    u, s, v = torch.svd(mat)
    vt = torch.transpose(v, 1, 2)
    det = torch.det(torch.matmul(u, vt)).view(-1, 1, 1)
    vt = torch.cat((vt[:, :2, :], vt[:, -1:, :] * det), 1)
    return torch.matmul(u, vt)


def test_project_so3_forward_and_backward():
    from chore_amd.recon.recon_fit_base import ReconFitterBase
    g = torch.Generator().manual_seed(0)
    M = torch.randn(64, 3, 3, generator=g)
    M[0] = torch.eye(3) + 1e-3 * torch.randn(3, 3, generator=g)          # near a rotation
    M[1] = -torch.eye(3) + 1e-2 * torch.randn(3, 3, generator=g)         # det < 0
    M[2] = torch.diag(torch.tensor([2.0, 1.0, 1e-6]))                    # nearly singular
    W = torch.randn(64, 3, 3, generator=g)
    Mc = M.clone().requires_grad_(True)
    Rc = ref_project_so3(Mc)
    (Rc * W).sum().backward()
    Mg = M.cuda().requires_grad_(True)
    Rg = ReconFitterBase.project_so3(Mg)
    (Rg * W.cuda()).sum().backward()
    R = Rg.detach().cpu()
    assert torch.allclose(R, Rc.detach(), atol=2e-6)
    assert torch.allclose(torch.bmm(R, R.transpose(1, 2)), torch.eye(3).expand(64, 3, 3), atol=1e-5)
    assert torch.allclose(torch.det(R), torch.ones(64), atol=1e-5)
    # gradients: skip the nearly singular sample, where torch.svd's backward itself is ill-conditioned
    ok = torch.ones(64, dtype=torch.bool); ok[2] = False
    gc, gg = Mc.grad[ok], Mg.grad.cpu()[ok]
    assert (gc - gg).abs().max() < 5e-4 * gc.abs().max()


from oracle.contact import contact_loss as ref_contact_loss  # noqa: E402  (restatement of recon_fit_base.py:553-608)


def test_contact_term_matches_restatement():
    """chore_contact_fwd/bwd against the ragged torch restatement: value and gradients, including a frame without
    any contact (skipped), a frame whose human side has no contact vertex (all vertices are used) and a part that
    occurs on one side only"""
    from chore_amd.recon.recon_fit_base import _ContactFn
    g = torch.Generator().manual_seed(3)
    B, Nh, No, P = 4, 700, 300, 14
    hum = (torch.randn(B, Nh, 3, generator=g) * 0.3).cuda()
    obj = (torch.randn(B, No, 3, generator=g) * 0.3 + 0.1).cuda()
    df_h = torch.rand(B, Nh, generator=g) * 0.3          # ~27 % below 0.08
    df_o = torch.rand(B, No, generator=g) * 0.3
    df_h[1] = 1.0; df_o[1] = 1.0                         # frame 1: no contact at all
    df_h[2] = 1.0                                        # frame 2: no human contact -> all vertices take part
    logits = torch.randn(B, P, No, generator=g)
    logits[:, 13] = -50.0                                # part 13 never predicted on the object
    labels = torch.randint(0, P, (Nh,), generator=g).cuda()
    df_h, df_o, logits = df_h.cuda(), df_o.cuda(), logits.cuda()
    h1, o1 = hum.clone().requires_grad_(True), obj.clone().requires_grad_(True)
    ref = ref_contact_loss(df_h, df_o, o1, h1, logits, labels)
    (ref * 3.0).backward()
    h2, o2 = hum.clone().requires_grad_(True), obj.clone().requires_grad_(True)
    got = _ContactFn.apply(h2, o2, df_h, df_o, logits, labels, 0.08)
    (got * 3.0).backward()
    assert abs(float(got) - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
    for a, b in ((h1.grad, h2.grad), (o1.grad, o2.grad)):
        assert (a - b).abs().max() < 1e-6 * max(1e-3, float(a.abs().max()))
    assert h2.grad[1].abs().max() == 0 and o2.grad[1].abs().max() == 0
    # nothing in contact anywhere: the reference omits the term, here it is exactly zero with zero gradients
    none = ref_contact_loss(df_h * 0 + 1, df_o * 0 + 1, obj, hum, logits, labels)
    z = _ContactFn.apply(h2, o2, df_h * 0 + 1, df_o * 0 + 1, logits, labels, 0.08)
    assert none is None and float(z) == 0.0


@pytest.fixture()
def fit_setup(opt):   # function scope: the tests optimise the parameters in place
    from chore_amd.lib_smpl.priors import synthetic_priors
    from chore_amd.lib_smpl.wrapper_pytorch import SMPLPyTorchWrapperBatch
    from chore_amd.model import CHORE
    from chore_amd.recon.recon_fit_behave import ReconFitterBehave
    from chore_amd.utils import synth
    from test_gpu_query import nhwc
    opt.compute_dtype = "fp32"
    B = 2
    net = CHORE(opt).cuda().eval()
    synth.load_synth_weights(net, seed=0)
    for p in net.parameters():
        p.requires_grad_(False)
    rs = np.random.RandomState(9)
    net.im_feat_list = [nhwc((rs.standard_normal((B, 256, 32, 32)) * 0.5).astype(np.float32))]
    net.tmpx = nhwc((rs.standard_normal((B, 64, 64, 64)) * 0.5).astype(np.float32))
    pose, betas, trans = synth.synth_smpl_params(B, seed=1)
    pose *= 0.3
    smpl = SMPLPyTorchWrapperBatch(synth.synth_smplh_model(0), B, betas=torch.from_numpy(betas),
                                   pose=torch.from_numpy(pose), trans=torch.from_numpy(trans)).cuda()
    body_prior, hand_prior = synthetic_priors(0)
    labels = torch.from_numpy(rs.randint(0, 14, 6890)).cuda()
    fitter = ReconFitterBehave.from_parts(device="cuda:0", part_labels=labels, body_prior=body_prior, hand_prior=hand_prior)
    cc = torch.tensor([synth.CROP_CENTER] * B).cuda()
    kpts = torch.from_numpy(np.concatenate([rs.uniform(100, 400, (B, 25, 2)), rs.uniform(0.2, 1, (B, 25, 1))], -1)
                            .astype(np.float32)).cuda()
    obj = torch.from_numpy((rs.standard_normal((B, 3000, 3)) * 0.15).astype(np.float32)).cuda()
    data = dict(net=net, query_dict={"crop_center": cc}, part_labels=labels.unsqueeze(0).repeat(B, 1),
                pose_init=torch.from_numpy(pose[:, 3:72]).cuda(), body_kpts=kpts, objects=obj, smpl=smpl,
                obj_R=torch.eye(3).repeat(B, 1, 1).cuda().requires_grad_(True),
                obj_t=torch.tensor([[0.2, 0.3, 2.3]] * B).cuda().requires_grad_(True),
                obj_s=torch.ones(B).cuda().requires_grad_(True))
    return fitter, net, smpl, data


def test_forward_smpl_loss_terms_and_descent(fit_setup):
    fitter, net, smpl, data = fit_setup
    split = fitter.split_smpl(smpl)
    ld = fitter.forward_smpl(split, data, "kpts")
    assert set(ld) == {"df_h", "pose", "hand", "part", "smplz", "pinit", "j2d"}
    wd = fitter.get_loss_weights()
    opt = torch.optim.Adam([split.trans, split.global_pose, split.body_pose, split.top_betas], 0.006)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = fitter.sum_dict(fitter.forward_smpl(split, data, "kpts"), wd, 1)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    for p in (split.trans, split.global_pose, split.body_pose, split.top_betas):
        assert torch.isfinite(p.grad).all() and p.grad.abs().max() > 0


@pytest.mark.parametrize("phase", ["kpts", "smpl all pose"])
def test_fused_loss_terms_equal_tensor_expressions(fit_setup, phase, monkeypatch):
    """the two operators of recon/fit_terms.py (chore_fit_smpl_terms / chore_fit_point_terms) against the tensor
    expressions they replace (the reference's own formulation, kept in ReconFitterBase): every term to 2e-6 relative,
    the gradient of every SMPL parameter w.r.t. every single term to 1e-5 of that gradient's largest entry"""
    from chore_amd.recon import fit_terms
    fitter, net, smpl, data = fit_setup
    split = fitter.split_smpl(smpl)
    params = {k: getattr(split, k) for k in ("trans", "global_pose", "body_pose", "hand_pose", "top_betas", "other_betas")}

    def run(torch_terms):
        monkeypatch.setattr(fit_terms, "TORCH_TERMS", torch_terms)
        ld = fitter.forward_smpl(split, data, phase)
        grads = {}
        for k, v in ld.items():
            gs = torch.autograd.grad(v, list(params.values()), retain_graph=True, allow_unused=True)
            grads[k] = {n: (None if g is None else g.detach().clone()) for n, g in zip(params, gs)}
        return {k: float(v.detach()) for k, v in ld.items()}, grads

    ref_v, ref_g = run(True)
    got_v, got_g = run(False)
    assert list(ref_v) == list(got_v)                       # same terms in the same (summation) order
    assert ("j2d" in got_v) == (phase == "kpts")
    for k in ref_v:
        assert abs(got_v[k] - ref_v[k]) <= 2e-6 * max(abs(ref_v[k]), 1e-3), (k, got_v[k], ref_v[k])
        for n in params:
            a, b = got_g[k][n], ref_g[k][n]
            if b is None or float(b.abs().max()) == 0.0:
                assert a is None or float(a.abs().max()) == 0.0, (k, n)
                continue
            assert a is not None, (k, n)
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), (k, n, float((a - b).abs().max()), float(b.abs().max()))


def test_forward_step_phases(fit_setup):
    fitter, net, smpl, data = fit_setup
    split = fitter.split_smpl(smpl)
    data = dict(data)
    data["smpl_center"] = fitter.compute_smpl_center_pred(data, net, smpl)
    R, t, s = data["obj_R"], data["obj_t"], data["obj_s"]
    ld = fitter.forward_step(net, split, data, R, t, s, "object only")
    assert set(ld) == {"object", "scale", "ocent"}
    ld = fitter.forward_step(net, split, data, R, t, s, "joint")
    assert {"object", "scale", "ocent"} <= set(ld)
    loss = fitter.sum_dict(ld, fitter.get_loss_weights(), 1)
    loss.backward()
    assert torch.isfinite(R.grad).all() and torch.isfinite(t.grad).all() and R.grad.abs().max() > 0


@pytest.mark.parametrize("phase", ["object only", "joint"])
def test_fused_object_terms_equal_tensor_expressions(fit_setup, phase, monkeypatch):
    """chore_fit_obj_transform / chore_fit_obj_terms / chore_fit_point_terms inside forward_step against the tensor
    expressions (bmm + broadcast adds, clamp / mse / mean): terms to 2e-6 relative, the gradients of obj_R, obj_t, obj_s
    w.r.t. every term to 2e-5 of that gradient's largest entry"""
    from chore_amd.recon import fit_terms
    fitter, net, smpl, data = fit_setup
    split = fitter.split_smpl(smpl)
    data = dict(data)
    data["smpl_center"] = fitter.compute_smpl_center_pred(data, net, smpl)
    rs = np.random.RandomState(3)
    R = (torch.eye(3).repeat(2, 1, 1) + torch.from_numpy(rs.standard_normal((2, 3, 3)).astype(np.float32)) * 0.05).cuda().requires_grad_(True)
    t = torch.tensor([[0.2, 0.3, 2.3], [0.1, 0.25, 2.4]]).cuda().requires_grad_(True)
    s = torch.tensor([1.05, 0.93]).cuda().requires_grad_(True)
    noise = torch.from_numpy(rs.uniform(0, 1, (2, 3, 3)).astype(np.float32)).cuda()

    def run(torch_terms):
        monkeypatch.setattr(fit_terms, "TORCH_TERMS", torch_terms)
        ld = fitter.forward_step(net, split, data, R, t, s, phase, noise=noise)
        grads = {k: torch.autograd.grad(v, [R, t, s], retain_graph=True, allow_unused=True) for k, v in ld.items()}
        return {k: float(v.detach()) for k, v in ld.items()}, grads

    ref_v, ref_g = run(True)
    got_v, got_g = run(False)
    assert list(ref_v) == list(got_v)
    for k in ref_v:
        assert abs(got_v[k] - ref_v[k]) <= 2e-6 * max(abs(ref_v[k]), 1e-3), (k, got_v[k], ref_v[k])
        for a, b, n in zip(got_g[k], ref_g[k], "Rts"):
            if b is None or float(b.abs().max()) == 0.0:
                assert a is None or float(a.abs().max()) == 0.0, (k, n)
                continue
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), (k, n, float((a - b).abs().max()), float(b.abs().max()))


def test_optimize_loops_run(fit_setup):
    fitter, net, smpl, data = fit_setup
    d = dict(data)
    smpl2, scale = fitter.optimize_smpl(smpl, d, iter_for_betas=1, iter_for_pose=1, iter_for_kpts=1, steps_per_iter=2, max_iter=1)
    assert scale.shape == (2,) and torch.isfinite(scale).all()
    d["smpl"] = smpl2
    out_smpl, R, t = fitter.optimize_smpl_object(net, d, obj_iter=1, joint_iter=1, steps_per_iter=2, max_iter=1)
    assert torch.isfinite(R).all() and torch.isfinite(t).all()


def _run_fit(opt, use_graphs):
    return run_fit_with(opt, use_graphs)


def run_fit_with(opt, use_graphs, silhouette=None, obj_iter=2, sil_iter=0, joint_iter=2):
    """both optimisation loops, short schedules, from identical initial state"""
    import copy
    from chore_amd.lib_smpl.priors import synthetic_priors
    from chore_amd.lib_smpl.wrapper_pytorch import SMPLPyTorchWrapperBatch
    from chore_amd.model import CHORE
    from chore_amd.recon.recon_fit_behave import ReconFitterBehave
    from chore_amd.utils import synth
    from test_gpu_query import nhwc
    opt = copy.copy(opt)
    opt.compute_dtype = "fp32"
    B = 2
    net = CHORE(opt).cuda().eval()
    synth.load_synth_weights(net, seed=0)
    for p in net.parameters():
        p.requires_grad_(False)
    rs = np.random.RandomState(9)
    net.im_feat_list = [nhwc((rs.standard_normal((B, 256, 32, 32)) * 0.5).astype(np.float32))]
    net.tmpx = nhwc((rs.standard_normal((B, 64, 64, 64)) * 0.5).astype(np.float32))
    pose, betas, trans = synth.synth_smpl_params(B, seed=1)
    pose *= 0.3
    smpl = SMPLPyTorchWrapperBatch(synth.synth_smplh_model(0), B, betas=torch.from_numpy(betas),
                                   pose=torch.from_numpy(pose), trans=torch.from_numpy(trans)).cuda()
    body_prior, hand_prior = synthetic_priors(0)
    labels = torch.from_numpy(rs.randint(0, 14, 6890)).cuda()
    fitter = ReconFitterBehave.from_parts(device="cuda:0", part_labels=labels, body_prior=body_prior, hand_prior=hand_prior)
    fitter.use_graphs = use_graphs
    fitter.adam_capturable = True   # the same Adam arithmetic in both runs
    cc = torch.tensor([synth.CROP_CENTER] * B).cuda()
    kpts = torch.from_numpy(np.concatenate([rs.uniform(100, 400, (B, 25, 2)), rs.uniform(0.2, 1, (B, 25, 1))], -1)
                            .astype(np.float32)).cuda()
    obj = torch.from_numpy((rs.standard_normal((B, 3000, 3)) * 0.15).astype(np.float32)).cuda()
    data = dict(net=net, query_dict={"crop_center": cc}, part_labels=labels.unsqueeze(0).repeat(B, 1),
                pose_init=torch.from_numpy(pose[:, 3:72]).cuda(), body_kpts=kpts, objects=obj, smpl=smpl,
                obj_R=torch.eye(3).repeat(B, 1, 1).cuda().requires_grad_(True),
                obj_t=torch.tensor([[0.2, 0.3, 2.3]] * B).cuda().requires_grad_(True),
                obj_s=torch.ones(B).cuda().requires_grad_(True))
    torch.manual_seed(11)   # CPU stream of the SO(3) perturbation
    smpl2, scale = fitter.optimize_smpl(smpl, data, iter_for_betas=2, iter_for_pose=2, iter_for_kpts=2, steps_per_iter=5,
                                        max_iter=1)
    data["smpl"] = smpl2
    if silhouette is not None:
        data["silhouette"] = silhouette
    _, R, t = fitter.optimize_smpl_object(net, data, obj_iter=obj_iter, joint_iter=joint_iter, steps_per_iter=5, max_iter=1,
                                          sil_iter=sil_iter)
    return [x.detach().cpu().numpy().copy() for x in (smpl2.pose, smpl2.betas, smpl2.trans, scale, R, t, data["obj_s"])]


def test_graph_replay_equals_eager(opt):
    """every inner step as a hipGraph replay (config 5) must fit the same parameters as issuing the ops one by one:
    same kernels in the same order with the same Adam arithmetic (capturable=True in both runs) -> equal to a few
    ulps"""
    eager = _run_fit(opt, False)
    graph = _run_fit(opt, True)
    for name, a, b in zip(("pose", "betas", "trans", "scale", "R", "t", "s"), eager, graph):
        assert np.isfinite(b).all(), name
        d = np.abs(a - b)
        assert d.max() < 1e-5, (name, d.max(), np.median(d))
    # and the run must have moved the parameters (the graph really executed)
    assert np.abs(graph[5] - np.array([[0.2, 0.3, 2.3]] * 2, np.float32)).max() > 1e-3


def test_fit_trajectories_match_reference(fit_setup):
    """10 Adam steps of forward_smpl('kpts') and forward_step('object only') against the trajectories the
    reference's own ReconFitterBehave produced on CPU (tests/golden/fit_trajectories.npz): per-step loss
    terms to 2e-3 relative (5e-5 absolute for terms that approach zero), fitted parameters: see `close`."""
    from conftest import golden
    fitter, net, smpl, data = fit_setup
    g = golden("fit_trajectories.npz")
    wd = fitter.get_loss_weights()
    # ---- SMPL, phase 'kpts', gradients accumulate over the steps (zero_grad once) ----
    split = fitter.split_smpl(smpl)
    opt = torch.optim.Adam([split.trans, split.global_pose, split.body_pose, split.top_betas, split.other_betas], 0.006)
    opt.zero_grad()
    keys = [str(k) for k in g["keys_a"]]
    for i in range(10):
        ld = fitter.forward_smpl(split, data, "kpts")
        got = np.array([float(ld[k].detach()) for k in keys])
        np.testing.assert_allclose(got, g["smpl_losses"][i], rtol=2e-3, atol=5e-5, err_msg=f"smpl step {i} {keys}")
        fitter.sum_dict(ld, wd, 1).backward()
        opt.step()
    def close(pairs, max_tol=1.2e-2):
        # Adam's update is ~lr*sign(g) while |g| >> sqrt(v): a component whose accumulated gradient passes
        # through zero takes a step (lr = 6e-3 here) in the other direction on fp32 round-off alone, in
        # any implementation (components with a near-zero gradient are steered by summation noise).  Measured
        # on the 158 SMPL components with two landmark products that differ in summation order only (library GEMM /
        # chore_landmarks_fwd): max 3.8e-3 / 7.6e-3, median 0.9e-4 / 1.2e-4, 31 / 29 components beyond 1e-3.  So: every
        # component within TWO steps, the median within 2e-4, three quarters within 1e-3 -- and, above, the loss terms
        # of all 10 steps within 2e-3, which is what the trajectories are optimised for.
        err = np.concatenate([np.abs(a - b).ravel() for a, b in pairs])
        assert err.max() < max_tol, (err.max(), np.sort(err)[-6:])
        assert np.median(err) < 2e-4, np.median(err)
        assert (err < 1e-3).mean() >= 0.75, (err < 1e-3).mean()

    close([(getattr(split, k).detach().cpu().numpy(), g["smpl_" + k])
           for k in ("trans", "global_pose", "body_pose", "top_betas", "other_betas")])
    # ---- object, phase 'object only' ----
    d = dict(data)
    d["smpl_center"] = torch.tensor([[0.0, 0.3, 2.2]] * 2).cuda()
    obj_R = torch.eye(3).repeat(2, 1, 1).cuda().requires_grad_(True)
    obj_t = torch.tensor([[0.2, 0.3, 2.3]] * 2).cuda().requires_grad_(True)
    obj_s = torch.ones(2).cuda().requires_grad_(True)
    opt = torch.optim.Adam([obj_t, obj_R, obj_s], lr=0.006)
    opt.zero_grad()
    torch.manual_seed(123)   # same CPU random stream for the 1e-4 noise of decopose_axis
    keys = [str(k) for k in g["keys_b"]]
    for i in range(10):
        ld = fitter.forward_step(net, split, d, obj_R, obj_t, obj_s, "object only")
        got = np.array([float(ld[k].detach()) for k in keys])
        np.testing.assert_allclose(got, g["obj_losses"][i], rtol=2e-3, atol=5e-5, err_msg=f"object step {i}")
        fitter.sum_dict(ld, wd, 1).backward()
        opt.step()
    # the raw 3x3 parameter obj_R drifts freely along the directions the SO(3) projection ignores (zero true
    # gradient, Adam amplifies round-off there -- in the reference too); what is fitted is the ROTATION
    def rot(m):
        u, _, vt = np.linalg.svd(m.astype(np.float64))
        d = np.linalg.det(u @ vt)
        return (u * np.stack([np.ones_like(d), np.ones_like(d), d], -1)[:, None, :]) @ vt

    close([(obj_t.detach().cpu().numpy(), g["obj_t"]), (obj_s.detach().cpu().numpy(), g["obj_s"]),
           (rot(obj_R.detach().cpu().numpy()), rot(g["obj_R"]))], max_tol=1.2e-2)


def test_coco_variant_matches_reference():
    """ReconFitterCoco (recon/recon_fit_coco.py:32-74): keypoint mapping with the mean crop centre and the loss weights,
    against values of the reference's own class (tests/golden/coco_fit.npz)"""
    from conftest import golden
    from chore_amd.recon.recon_fit_coco import ReconFitterCoco
    g = golden("coco_fit.npz")
    f = ReconFitterCoco.from_parts(device="cuda:0")
    t = lambda k: torch.from_numpy(g[k]).cuda()     # noqa: E731
    out = f.scale_body_kpts(t("kpts"), t("resize_scale"), t("crop_scale"), t("old_crop_center"))
    np.testing.assert_allclose(out.cpu().numpy(), g["kpts_out"], rtol=1e-6, atol=1e-3)
    wd = f.get_loss_weights()
    assert sorted(wd) == [str(k) for k in g["weight_names"]]
    for k, v in zip(g["weight_names"], g["weights_at_2_3"]):
        assert abs(float(wd[str(k)](2.0, 3)) - float(v)) <= 1e-12 * abs(float(v))


def test_fused_adam_step_equals_torch_adam():
    """chore_fit_adam_step + chore_fit_stop_rule (csrc/fit_step.hip: all tensors of a step in one launch, the stop rule and
    the step counter in another) against torch.optim.Adam(capturable=True) and the tensor-op stop rule of EagerStep, 30
    steps on random gradients with the rule firing in between: parameters, moments, flags"""
    import os
    from chore_amd.recon import graph_step as gs
    torch.manual_seed(11)
    shapes = [(2, 72), (2, 10), (2, 3), (2, 3, 3), (2,)]
    res = {}
    for mode in ("fused", "torch"):
        torch.manual_seed(12)
        params = [torch.randn(s, device="cuda").requires_grad_(True) for s in shapes]
        targets = [torch.randn(s, device="cuda") for s in shapes]
        if mode == "torch":
            os.environ["CHORE_FIT_TORCH_ADAM"] = "1"
        try:
            prev = torch.tensor(300.0, device="cuda")
            calls = {"n": 0}

            def loss_fn(decay):
                calls["n"] += 1
                k = calls["n"]
                # a loss whose value stalls for a while (the rule fires), gradients that keep changing
                base = sum(((p - t) ** 2).sum() for p, t in zip(params, targets))
                return base * (0.0 if 12 <= k <= 14 else 1.0) + 5.0 + 0 * (1 + decay)
            st = gs.EagerStep(params, 0.02, loss_fn, 0.001, prev, capturable=True)
            assert isinstance(st.opt, gs.FusedAdam) == (mode == "fused")
            trace = []
            for it in range(6):
                st.begin_outer(it, armed=it >= 2, zero=True)
                for _ in range(5):
                    st.step()
                    trace.append((float(st.loss), bool(st.stop), float(st.prev)))
        finally:
            os.environ.pop("CHORE_FIT_TORCH_ADAM", None)
        res[mode] = ([p.detach().clone() for p in params], trace)
    (pa, ta), (pb, tb) = res["fused"], res["torch"]
    assert [t[1] for t in ta] == [t[1] for t in tb] and any(t[1] for t in ta)          # the latch fires at the same step
    for (la, _, pva), (lb, _, pvb) in zip(ta, tb):
        assert abs(la - lb) <= 2e-6 * abs(lb) and abs(pva - pvb) <= 2e-6 * abs(pvb)
    for a, b in zip(pa, pb):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())


def test_stop_rule_inside_the_sum_launch_equals_the_separate_launches():
    """chore_fit_weighted_sum_step (sum of the loss terms + its backward for the stepper's seed + the stop rule, one launch,
    the rule BEFORE the step's Adam) against the three separate launches (rule after Adam): 30 steps with the rule firing
    in between -- losses, flags, previous loss, step counter, parameters and accumulated gradients bit for bit, eager and as a
    hipGraph replay; a loss_fn that changes the sum is refused."""
    from chore_amd.recon import graph_step as gs
    shapes = [(2, 72), (2, 10), (2, 3), (2, 3, 3), (2,)]
    coeffs = [1.0, 0.5, 2.0, 0.25, 3.0, 1.0]
    res = {}
    for mode in ("split", "fused", "fused graph"):
        torch.manual_seed(12)
        params = [torch.randn(s, device="cuda").requires_grad_(True) for s in shapes]
        targets = [torch.randn(s, device="cuda") for s in shapes]
        five = torch.tensor(5.0, device="cuda")
        gate = torch.ones((), device="cuda")          # 0 for a few steps: the loss stalls, the rule fires
        prev = torch.tensor(300.0, device="cuda")

        def loss_fn(decay):
            terms = [((p - t) ** 2).sum() * gate for p, t in zip(params, targets)] + [five]
            return gs.weighted_sum(terms, coeffs, 1 + decay)
        cls = gs.GraphedStep if mode == "fused graph" else gs.EagerStep
        st = cls(params, 0.02, loss_fn, 0.001, prev, capturable=True, fuse_rule=mode != "split")
        trace, k = [], 0
        for it in range(6):
            st.begin_outer(it, armed=it >= 2, zero=True)
            for _ in range(5):
                k += 1
                gate.fill_(0.0 if 12 <= k <= 14 else 1.0)
                st.step()
                trace.append((float(st.loss), bool(st.stop), float(st.prev), float(st.opt.step_t)))
        res[mode] = (trace, [p.detach().clone() for p in params], [p.grad.clone() for p in params])
    assert any(t[1] for t in res["split"][0]) and not res["split"][0][10][1]
    for mode in ("fused", "fused graph"):
        assert res[mode][0] == res["split"][0], mode
        for a, b in zip(res[mode][1] + res[mode][2], res["split"][1] + res["split"][2]):
            assert torch.equal(a, b), mode
    params = [torch.randn(3, device="cuda").requires_grad_(True)]
    st = gs.EagerStep(params, 0.02, lambda decay: 2.0 * gs.weighted_sum([(params[0] ** 2).sum()], [1.0], 1 + decay), 0.001,
                      torch.tensor(300.0, device="cuda"), capturable=True, fuse_rule=True)
    with pytest.raises(RuntimeError, match="fuse_rule=False"):
        st.step()


def test_fused_adam_on_column_slices_of_a_batch_equals_torch_adam():
    """the split SMPL parameters of a multi-frame batch are column slices of the wrapper's storage (b[:, :2], p[:, 3:66], ...:
    non-contiguous for B > 1, lib_smpl/wrapper_pytorch.py from_smpl): the accumulate-in-Adam launch updates them in place
    through their row stride -- same trajectory as torch.optim.Adam(capturable=True) on the same views, the storage they
    alias written through, gradients accumulating over the inner steps of an outer iteration"""
    import os
    from chore_amd.recon import graph_step as gs
    res = {}
    for mode in ("fused", "torch"):
        torch.manual_seed(5)
        B = 3
        pose, betas = torch.randn(B, 72, device="cuda"), torch.randn(B, 10, device="cuda")
        trans = torch.randn(B, 3, device="cuda")
        views = [betas[:, :2], betas[:, 2:], pose[:, :3], pose[:, 3:66], trans]
        assert not views[0].is_contiguous()
        params = [torch.nn.Parameter(v) for v in views]             # like the wrapper: parameters that alias the storage
        targets = [torch.randn(v.shape, device="cuda") for v in views]
        if mode == "torch":
            os.environ["CHORE_FIT_TORCH_ADAM"] = "1"
        try:
            prev = torch.tensor(300.0, device="cuda")

            def loss_fn(decay):
                return sum(((p - t) ** 2).sum() * (k + 1) for k, (p, t) in enumerate(zip(params, targets))) / (1 + decay)
            st = gs.EagerStep(params[:4], 0.02, loss_fn, 0.001, prev, capturable=True, carry=[params[4]])
            assert isinstance(st.opt, gs.FusedAdam) == (mode == "fused")
            for it in range(4):
                st.begin_outer(it, armed=False, zero=True)
                for _ in range(3):
                    st.step()
        finally:
            os.environ.pop("CHORE_FIT_TORCH_ADAM", None)
        res[mode] = (pose.clone(), betas.clone(), trans.clone(), [p.grad.clone() for p in params])
    for a, b in zip(res["fused"][:3], res["torch"][:3]):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    assert float((res["fused"][0][:, 66:] - res["torch"][0][:, 66:]).abs().max()) == 0       # the hand pose columns: nobody's
    for a, b in zip(res["fused"][3], res["torch"][3]):                                          # accumulated gradients, the carried leaf too
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())


def test_rot_noise_operator():
    """chore_fit_rot_noise: rot + 1e-4 * noise[k] bit for bit as the tensor expression, the device counter advanced by one per
    call, the gradient passed through to rot"""
    from chore_amd.recon import fit_terms
    rs = np.random.RandomState(2)
    B, S = 3, 5
    rot = torch.from_numpy(rs.standard_normal((B, 3, 3)).astype(np.float32)).cuda().requires_grad_(True)
    noise = torch.from_numpy(rs.uniform(0, 1, (S, B, 3, 3)).astype(np.float32)).cuda()
    k = torch.zeros(1, dtype=torch.long, device="cuda")
    assert fit_terms.rot_noise_supported(rot, noise, k)
    for step in range(S):
        out = fit_terms.rot_noise(rot, noise, k)
        assert int(k) == step + 1
        assert torch.equal(out.detach(), rot.detach() + 1e-4 * noise[step])
    w = torch.from_numpy(rs.standard_normal((B, 3, 3)).astype(np.float32)).cuda()
    k.zero_()
    (fit_terms.rot_noise(rot, noise, k) * w).sum().backward()
    assert torch.equal(rot.grad, w)
