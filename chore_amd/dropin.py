"""Run the reference's own scripts on this implementation without editing them.

    python -m chore_amd.dropin demo.py chore-release -s example/...          (from the root of a CHORE checkout)
    python -m chore_amd.dropin recon/recon_fit_behave.py chore-release -sn test -s <sequence>
    python -m torch.distributed.run --nproc-per-node 8 -m chore_amd.dropin train_launch.py -en chore-release

`install()` registers the hot-path modules of this package under the import names the reference's scripts use
(`from model import CHORE` /root/reference/model/__init__.py:1, `from recon.generator import Generator`,
`from recon.recon_fit_base import ReconFitterBase`, `from lib_smpl.wrapper_pytorch import ...`), so that

    model, model.chore, model.camera, model.geometry
    recon, recon.generator, recon.recon_fit_base, recon.recon_fit_behave, recon.recon_fit_coco, recon.obj_pose_roi,
    recon.eval.chamfer_distance, recon.eval.pose_utils
    lib_smpl, lib_smpl.const, lib_smpl.wrapper_pytorch, lib_smpl.smpl_generator

resolve to chore_amd's classes, while every submodule this package does not replace (recon.opt_utils, recon.bbox,
lib_smpl.body_landmark, data.*, behave.*, config.*, trainer.*, utils.*) is still found in the reference checkout: the
alias packages get the checkout's directories appended to their search path.  Nothing is copied and no file of the
reference is modified.
"""
import importlib
import os
import runpy
import sys

ALIASES = {
    "model": "chore_amd.model",
    "model.chore": "chore_amd.model.chore",
    "model.camera": "chore_amd.model.camera",
    "model.geometry": "chore_amd.model.geometry",
    "recon": "chore_amd.recon",
    "recon.generator": "chore_amd.recon.generator",
    "recon.recon_fit_base": "chore_amd.recon.recon_fit_base",
    "recon.recon_fit_behave": "chore_amd.recon.recon_fit_behave",
    "recon.recon_fit_coco": "chore_amd.recon.recon_fit_coco",
    "recon.obj_pose_roi": "chore_amd.recon.obj_pose_roi",
    "recon.eval": "chore_amd.recon.eval",
    "recon.eval.chamfer_distance": "chore_amd.recon.eval.chamfer_distance",
    "recon.eval.pose_utils": "chore_amd.recon.eval.pose_utils",
    "lib_smpl": "chore_amd.lib_smpl",
    "lib_smpl.const": "chore_amd.lib_smpl.const",
    "lib_smpl.wrapper_pytorch": "chore_amd.lib_smpl.wrapper_pytorch",
    "lib_smpl.smpl_generator": "chore_amd.lib_smpl.smpl_generator",
}


def install(reference_root=None):
    """alias the modules; `reference_root` (default: the working directory) is where the rest of CHORE lives"""
    root = os.path.abspath(reference_root or os.getcwd())
    for name, target in ALIASES.items():
        mod = importlib.import_module(target)
        sys.modules[name] = mod
        sub = os.path.join(root, *name.split("."))
        if hasattr(mod, "__path__") and os.path.isdir(sub) and sub not in mod.__path__:
            mod.__path__.append(sub)       # submodules we do not provide are the checkout's own files
    if root not in sys.path:
        sys.path.insert(0, root)
    return root


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    install()
    script = argv[0]
    sys.argv = argv
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
