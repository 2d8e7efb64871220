"""CPU: the checkpoint format of the reference's Trainer round-trips through `Generator.load_checkpoint`.

Writer side restated from /root/reference/trainer/trainer.py:186-206 (`save_checkpoint`: file name
`checkpoint_{h}h:{m}m:{s}s_{training_time}.tar`, keys `training_time, epoch, model_state_dict, optimizer_state_dict`; under DDP
the state dict carries a `module.` prefix) and :311-315 (`update_vmin_file`: `np.save(exp_path + 'val_min={epoch}',
[epoch, val_loss, ck_file])`).  Reader side is chore_amd/recon/generator.py (= recon/generator.py:219-267: newest by training
time unless a `val_min=*` file names the best one; `module.` stripped when `multi_gpus`)."""
import os

import numpy as np
import pytest
import torch

from chore_amd.model import CHORE
from chore_amd.recon.generator import Generator


def convert_secs(sec):      # trainer/trainer.py:345-349
    return int(sec / 3600), int((sec / 60) % 60), int(sec % 60)


def save_checkpoint(ck_dir, model_sd, optimizer, epoch, training_time):
    name = "checkpoint_{}h:{}m:{}s_{}.tar".format(*[*convert_secs(training_time), training_time])
    torch.save({"training_time": training_time, "epoch": epoch, "model_state_dict": model_sd,
                "optimizer_state_dict": optimizer.state_dict()}, os.path.join(ck_dir, name))
    return name


@pytest.fixture(scope="module")
def nets(opt):
    torch.manual_seed(3)
    a, b = CHORE(opt), CHORE(opt)
    with torch.no_grad():
        for i, net in enumerate((a, b)):
            for p in net.parameters():
                p.add_(0.01 * (i + 1) * torch.randn_like(p))
    return a, b


def _equal(net, sd):
    got = net.state_dict()
    assert set(got) == set(sd)
    return all(torch.equal(got[k], sd[k]) for k in sd)


@pytest.mark.parametrize("ddp_prefix", [True, False])
def test_checkpoint_roundtrip_newest_then_val_min_then_named(opt, nets, tmp_path, ddp_prefix):
    a, b = nets
    exp = tmp_path / "experiments" / "chore-release"
    ck_dir = exp / "checkpoints"
    ck_dir.mkdir(parents=True)
    pre = (lambda sd: {"module." + k: v for k, v in sd.items()}) if ddp_prefix else (lambda sd: dict(sd))
    opt_a = torch.optim.Adam(a.parameters(), lr=1e-4)
    name_a = save_checkpoint(str(ck_dir), pre(a.state_dict()), opt_a, epoch=3, training_time=7384.25)      # 2h:3m:4s
    name_b = save_checkpoint(str(ck_dir), pre(b.state_dict()), opt_a, epoch=7, training_time=36125.5)      # 10h:2m:5s
    assert name_a == "checkpoint_2h:3m:4s_7384.25.tar" and name_b == "checkpoint_10h:2m:5s_36125.5.tar"

    def load(checkpoint=None):
        net = CHORE(opt)
        gen = Generator(net, "chore-release", threshold=2.0, checkpoint=checkpoint, device=torch.device("cpu"),
                        multi_gpus=ddp_prefix, checkpoint_root=str(tmp_path / "experiments"))
        assert all(not p.requires_grad for p in gen.model.parameters())      # generator.py:46-47
        return net, gen

    # no val_min file: the checkpoint with the largest training time (numeric, not lexicographic: 36125.5 > 7384.25)
    net, gen = load()
    assert _equal(net, b.state_dict()) and not _equal(net, a.state_dict())
    assert gen.load_checkpoint(None) == (7, 36125.5)
    # a val_min file names the best checkpoint (written like trainer.py:311-315: a string array, np.save appends .npy)
    np.save(str(exp / "val_min=3"), [3, 0.125, name_a])
    net, gen = load()
    assert _equal(net, a.state_dict())
    assert gen.load_checkpoint(None) == (3, 7384.25)
    # ... unless the file it names is gone: back to the newest (generator.py:224-226)
    os.remove(str(exp / "val_min=3.npy"))
    np.save(str(exp / "val_min=5"), [5, 0.1, "checkpoint_0h:0m:1s_1.0.tar"])
    net, _ = load()
    assert _equal(net, b.state_dict())
    # an explicit name wins over both
    net, gen = load(checkpoint=name_a)
    assert _equal(net, a.state_dict())


def test_ddp_checkpoint_without_multi_gpus_is_rejected(opt, nets, tmp_path):
    """a DDP checkpoint (`module.` keys) read with multi_gpus=False fails loudly in load_state_dict, like the reference"""
    a, _ = nets
    ck_dir = tmp_path / "experiments" / "e" / "checkpoints"
    ck_dir.mkdir(parents=True)
    save_checkpoint(str(ck_dir), {"module." + k: v for k, v in a.state_dict().items()}, torch.optim.Adam(a.parameters()), 1, 10.0)
    with pytest.raises(RuntimeError):
        Generator(CHORE(opt), "e", device=torch.device("cpu"), multi_gpus=False, checkpoint_root=str(tmp_path / "experiments"))


def test_no_checkpoints(opt, tmp_path, capsys):
    (tmp_path / "experiments" / "e" / "checkpoints").mkdir(parents=True)
    gen = Generator(CHORE(opt), "e", device=torch.device("cpu"), checkpoint_root=str(tmp_path / "experiments"))
    assert gen.load_checkpoint(None) == (0, 0)
    assert "No checkpoints found" in capsys.readouterr().out
