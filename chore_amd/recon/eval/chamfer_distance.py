"""Chamfer distance between two point clouds on the GPU (fp64).

Counterpart of /root/reference/recon/eval/chamfer_distance.py:10-52 (sklearn kd-tree nearest neighbours): same
signature and return value -- the mean EUCLIDEAN nearest-neighbour distance of the requested direction(s), 'bi' adding
the two -- computed by chore_eval_chamfer (csrc/eval_metrics.hip: exhaustive tiled search, fixed-order reductions)."""
import numpy as np
import torch

from ... import _lib


def _dev64(a, dev):
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64)), device=dev)


def chamfer_distance(x, y, metric="l2", direction="bi", device="cuda:0"):
    if metric != "l2":
        raise ValueError("the GPU implementation provides the l2 metric the reference evaluates with")
    if direction not in ("bi", "x_to_y", "y_to_x"):
        raise ValueError("Invalid direction type. Supported types: 'y_x', 'x_y', 'bi'")
    dev = torch.device(device)
    h = _lib.handle(dev.index or 0)
    xd, yd = _dev64(x, dev), _dev64(y, dev)
    if xd.dim() != 2 or xd.shape[1] != 3 or yd.dim() != 2 or yd.shape[1] != 3:
        raise ValueError("expected [n_points, 3] arrays")
    out = torch.empty(2, dtype=torch.float64, device=dev)
    ws = torch.empty(_lib.lib.chore_eval_chamfer_workspace_bytes(xd.shape[0], yd.shape[0]), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib.chore_eval_chamfer(h, xd.data_ptr(), xd.shape[0], yd.data_ptr(), yd.shape[0], out.data_ptr(),
                                           ws.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), h, "chore_eval_chamfer")
    x_to_y, y_to_x = out.cpu().tolist()
    return {"bi": x_to_y + y_to_x, "x_to_y": x_to_y, "y_to_x": y_to_x}[direction]
