"""GPU: the DDP training step (train_launch.py:30 + trainer/trainer.py:76-131) with two ranks.

The box has one GPU, so both ranks share cuda:0 and the gradient all-reduce goes over gloo (the production launch,
bench.py --mode train, uses backend "nccl" = RCCL with one GPU per rank); what is checked is that torch's DDP reducer sees
the gradients our HIP autograd nodes produce -- find_unused_parameters=True like the reference, because the bn4
affines of the blocks without downsample never receive one -- and that what it leaves in .grad is the average of the
two ranks' own gradients (each recomputed here in one process on that rank's sample).  It is NOT compared with the
gradient of the two-sample batch: the reference's loss couples the samples of a batch (the (B,1,1,N) mask of
model/chore.py:214-220 broadcasts against the (B,3,N) centre error into a (B,B,3,N) tensor), so a DDP step of the
reference does not equal its single-process step on the concatenated batch either."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _batch(keys, g, sl):
    return {k: torch.from_numpy(g[k][sl]).cuda() for k in keys}


def _worker(rank, world, port, out_path):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    from conftest import golden
    from test_gpu_encoder import make_net
    import argparse
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    opt = argparse.Namespace(input_type="RGBM3", norm="group", num_stack=5, num_hourglass=2, hg_down="ave_pool",
                             hourglass_dim=256, skip_hourglass=True, z_feat="xyz", projection_mode="perspective",
                             loadSize=1200, net_img_size=[512, 512], gpu_id=0)
    net = make_net(opt, "fp32")
    net.train(True)
    for p in net.parameters():
        p.requires_grad_(True)
    ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0], find_unused_parameters=True)
    g = golden("train_loss.npz")
    keys = ("images", "points", "df_h", "df_o", "parts_gt", "pca_gt", "body_center", "obj_center", "crop_center")
    error, _ = ddp(**_batch(keys, g, slice(rank, rank + 1)))
    error.backward()
    if rank == 0:
        names = ["image_filter.conv2.conv1.weight", "image_filter.m2.b2_plus_1.bn2.weight", "image_filter.al3.bias",
                 "df.0.weight", "center_predictor.6.bias", "image_filter.conv1.weight"]
        params = dict(net.named_parameters())
        np.savez(out_path, **{n: params[n].grad.detach().cpu().numpy() for n in names},
                 unused=np.array([params["image_filter.m0.b1_2.bn4.weight"].grad is None or
                                  float(params["image_filter.m0.b1_2.bn4.weight"].grad.abs().max()) == 0.0]))
    dist.destroy_process_group()


def test_ddp_two_ranks_average_equals_full_batch(opt, tmp_path):
    import copy
    import torch.multiprocessing as mp
    from conftest import golden
    from test_gpu_encoder import make_net
    out = str(tmp_path / "ddp.npz")
    mp.spawn(_worker, args=(2, 29571, out), nprocs=2, join=True)
    got = np.load(out)
    assert bool(got["unused"][0])
    g = golden("train_loss.npz")
    keys = ("images", "points", "df_h", "df_o", "parts_gt", "pca_gt", "body_center", "obj_center", "crop_center")
    per_rank = []
    for r in range(2):
        net = make_net(copy.copy(opt), "fp32")
        net.train(True)
        for p in net.parameters():
            p.requires_grad_(True)
        error, _ = net.forward(**_batch(keys, g, slice(r, r + 1)))
        error.backward()
        params = dict(net.named_parameters())
        per_rank.append({n: params[n].grad.detach().cpu().numpy() for n in got.files if n != "unused"})
    for n in per_rank[0]:
        ref = (per_rank[0][n] + per_rank[1][n]) / 2
        assert np.abs(got[n] - ref).max() < 2e-5 * np.abs(ref).max(), (n, np.abs(got[n] - ref).max(), np.abs(ref).max())
