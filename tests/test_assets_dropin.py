"""CPU: the file readers behind FileAssets and the import aliases of chore_amd.dropin.

Where /root/reference exists (the build container) the readers are checked against the reference's own asset files and
the part labels its own loader returned (tests/golden/fit_init.npz); on the GPU box those checks are skipped -- the
reference does not travel."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import golden

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
have_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "assets")), reason="reference checkout not present")


def test_ply_roundtrip(tmp_path):
    from chore_amd.recon.assets import read_ply
    from chore_amd.recon.recon_fit_base import write_ply
    from meshes import icosphere
    v, f = icosphere(2)
    p = str(tmp_path / "m.ply")
    write_ply(p, v, f)
    v2, f2 = read_ply(p)
    assert np.allclose(v2, v.astype(np.float32)) and np.array_equal(f2, f)
    # ascii variant
    with open(p, "w") as fh:
        fh.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(v), len(f)))
        for r in v:
            fh.write("%r %r %r\n" % tuple(float(x) for x in r))
        for r in f:
            fh.write("3 %d %d %d\n" % tuple(r))
    v3, f3 = read_ply(p)
    assert np.allclose(v3, v) and np.array_equal(f3, f)


def test_surface_sampling_is_on_the_mesh():
    from chore_amd.recon.recon_fit_base import sample_surface
    from meshes import icosphere
    v, f = icosphere(3)
    pts = sample_surface(v, f, 3000, np.random.RandomState(0))
    r = np.linalg.norm(pts, axis=1)
    assert pts.shape == (3000, 3) and r.max() <= 1.0 + 1e-9 and r.min() > 0.98      # inside the facets of a unit sphere
    assert np.abs(pts.mean(0)).max() < 0.05                                         # uniform over the surface


def test_chumpy_free_smpl_reader(tmp_path):
    """the SMPL model reader on an .npz and on a pickle that holds 'chumpy' objects (unpickled through the shim)"""
    import pickle
    import types
    from chore_amd.recon.assets import load_smpl_model
    from chore_amd.utils import synth
    m = synth.synth_smplh_surface_model(0)
    kt = np.stack([np.asarray(m["parents"]), np.arange(52)])
    raw = dict(v_template=m["v_template"], shapedirs=m["shapedirs"], posedirs=m["posedirs"], J_regressor=m["J_regressor"],
               weights=m["weights"], kintree_table=kt, f=m["f"])
    np.savez(tmp_path / "SMPLH_male.npz", **raw)
    a = load_smpl_model(str(tmp_path / "SMPLH_male.npz"))
    assert np.array_equal(a["posedirs"], m["posedirs"]) and a["parents"] == [int(p) for p in m["parents"]]
    # a stand-in 'chumpy' package so that pickling records chumpy.ch.Ch as the class of the arrays
    ch = types.ModuleType("chumpy.ch")

    class Ch:
        def __init__(self, x):
            self.x = x
    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    ch.Ch = Ch
    pkg = types.ModuleType("chumpy")
    pkg.ch = ch
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = pkg, ch
    try:
        import scipy.sparse as sp
        raw2 = dict(raw, v_template=Ch(raw["v_template"]), shapedirs=Ch(raw["shapedirs"]), posedirs=Ch(raw["posedirs"]),
                    weights=Ch(raw["weights"]), J_regressor=sp.csc_matrix(raw["J_regressor"]))
        with open(tmp_path / "SMPLH_female.pkl", "wb") as fh:
            pickle.dump(raw2, fh)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
    b = load_smpl_model(str(tmp_path / "SMPLH_female.pkl"))
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights"):
        assert np.array_equal(b[k], a[k]), k


@have_ref
def test_file_assets_read_the_reference_asset_files():
    from chore_amd.recon.assets import FileAssets
    fa = FileAssets(os.path.join(REF, "assets"))
    labels = fa.part_labels()
    assert np.array_equal(labels, golden("fit_init.npz")["part_labels"].astype(np.int32))   # = the reference loader's output
    b25, face, hand = fa.regressors()
    assert b25.shape == (25, 6890) and face.shape == (70, 6890) and hand.shape == (42, 6890)
    assert np.allclose(b25.sum(1), 1.0, atol=1e-4)
    assert np.allclose(fa.mean_hand_pose(), golden("fit_init.npz")["mean_hand_pose"])
    import pickle
    dat = pickle.load(open(os.path.join(REF, "assets/priors/body_prior.pkl"), "rb"))
    assert np.asarray(dat["mean"]).shape == (63,) and np.asarray(dat["precision"]).shape == (63, 63)


@have_ref
def test_dropin_aliases_resolve_from_a_reference_checkout():
    """`from model import CHORE` etc. inside the reference tree resolve to this package; modules it does not replace
    still come from the checkout (config.config_loader)"""
    code = ("import chore_amd.dropin as d; d.install(); "
            "from model import CHORE; from model.camera import KinectColorCamera; from recon.generator import Generator; "
            "from recon.recon_fit_base import ReconFitterBase, RECON_PATH, BEHAVE_PATH, SMPL_ASSETS_ROOT; "
            "from recon.recon_fit_behave import ReconFitterBehave, recon_fit, RECON_PATH as RP2; "
            "from recon.recon_fit_coco import ReconFitterCoco; "
            "from lib_smpl.const import SMPL_POSE_PRAMS_NUM; "
            "from lib_smpl.wrapper_pytorch import SMPLPyTorchWrapperBatch, SMPL_MODEL_ROOT, SMPL_ASSETS_ROOT as SA2; "
            "import inspect; assert RECON_PATH == RP2 and RECON_PATH is not None and SMPL_ASSETS_ROOT == 'assets' == SA2; "
            "assert inspect.signature(ReconFitterBase.__init__).parameters['outpath'].default == RECON_PATH; "
            "import config.config_loader as cl; "
            "print(CHORE.__module__, Generator.__module__, ReconFitterBase.__module__, cl.__file__)")
    env = dict(os.environ, PYTHONPATH=REPO)
    out = subprocess.run([sys.executable, "-c", code], cwd=REF, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    mods = out.stdout.split()
    assert mods[0] == "chore_amd.model.chore" and mods[1] == "chore_amd.recon.generator"
    assert mods[2] == "chore_amd.recon.recon_fit_base" and mods[3].startswith(REF)
