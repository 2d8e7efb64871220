"""The reference's path constants (./PATHS.yml of a CHORE checkout: recon/recon_fit_base.py:39-45, lib_smpl/wrapper_pytorch.py:16-19).
The reference reads the file at import time and fails without it; here a missing file or key gives None, so the package also
imports outside a checkout (the device path never needs the paths)."""
import os


def load(root=None):
    p = os.path.join(root or os.getcwd(), "PATHS.yml")
    if not os.path.isfile(p):
        return {}
    try:
        import yaml
        with open(p, "r") as f:
            return yaml.safe_load(f) or {}
    except Exception:
        return {}


_P = load()
CODE = _P.get("CODE")
BEHAVE_PATH = _P.get("BEHAVE_PATH")
PROCESSED_PATH = _P.get("PROCESSED_PATH")
RECON_PATH = _P.get("RECON_PATH")
SMPL_ASSETS_ROOT = _P.get("SMPL_ASSETS_ROOT", "assets")
SMPL_MODEL_ROOT = _P.get("SMPL_MODEL_ROOT")
