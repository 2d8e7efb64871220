from .smpl_layer import SMPL_Layer  # noqa: F401
