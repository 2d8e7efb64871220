"""ORACLE (test infrastructure only -- never imported by chore_amd/).   *** PARITY UNPINNED at the pytorch3d boundary ***

torch (CPU or any device) restatement of the contact term of the joint fit:
  ReconFitterBase.compute_contact_loss      /root/reference/recon/recon_fit_base.py:553-608
      masks df < 0.08 on both sides, per frame the contact vertices (all vertices of a side that has none), paired by
      part label i = 0..13 (SMPL part labels on the human side, argmax of the predicted part logits on the object side),
      every non-empty pair of clouds appended to two lists
  pytorch3d.loss.chamfer_distance(Pointclouds(list_h), Pointclouds(list_o))      :605-607

pytorch3d is a third-party dependency that is neither vendored in /root/reference nor pinned (requirements.txt:22
installs its master branch from git) nor installed here, and the reference holds no test or vector for the term.  Its
PUBLISHED definition (docs of pytorch3d.loss.chamfer_distance, defaults: squared L2 nearest neighbour,
point_reduction="mean", batch_reduction="mean", the two directions added; ragged clouds keep their own lengths) is
restated in `Pointclouds` / `chamfer_distance` below.  The pairing loop itself IS pinned: tests/golden/make_golden.py
runs the reference's own compute_contact_loss with these two stand-ins injected for the missing package
(tests/golden/fit_schedule.npz, joint phase).
"""
import torch


class Pointclouds:
    """holder of a list of (N_i, 3) clouds (the only use the reference makes of pytorch3d.structures.Pointclouds)"""

    def __init__(self, points):
        self.points = list(points)


def chamfer_distance(x, y):
    """-> (sum over both directions of [mean over clouds of (mean over the cloud's points of the squared distance to
    the nearest point of the paired cloud)], None)"""
    da, db = [], []
    for a, b in zip(x.points, y.points):
        d = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
        da.append(d.min(1)[0].mean())
        db.append(d.min(0)[0].mean())
    return torch.stack(da).mean() + torch.stack(db).mean(), None


def contact_loss(df_hum_o, df_obj_h, obj, hum, part_logits, part_labels, n_parts=14, thres=0.08):
    """the whole term; returns None where the reference leaves `contact` out of the loss dict (no pair found)"""
    mask_o, mask_h = df_obj_h < thres, df_hum_o < thres
    part_o = torch.argmax(part_logits, 1)
    ch_list, co_list = [], []
    for hv, ov, mh, mo, po in zip(hum, obj, mask_h, mask_o, part_o):
        ch, co = int(mh.sum()), int(mo.sum())
        if ch + co == 0:
            continue
        obj_v, label_o = (ov[mo], po[mo]) if co > 0 else (ov, po)
        hum_v, label_h = (hv[mh], part_labels[mh]) if ch > 0 else (hv, part_labels)
        for i in range(n_parts):
            hi, oi = torch.where(label_h == i)[0], torch.where(label_o == i)[0]
            if hi.numel() == 0 or oi.numel() == 0:
                continue
            ch_list.append(hum_v[hi])
            co_list.append(obj_v[oi])
    if not ch_list:
        return None
    return chamfer_distance(Pointclouds(ch_list), Pointclouds(co_list))[0]
