"""chore_amd -- MI355X-native (gfx950) implementation of the CHORE field-query and
SMPL+object fitting hot path.

The package keeps the reference's Python API surface for the hot path
(`model.CHORE` at /root/reference/model/chore.py:10, `recon.recon_fit_base` helpers) and routes
every compute step through the C-ABI library `libchore_hip.so` (include/chore_hip.h), which holds
the hand-written HIP kernels.  There is NO CPU fallback: importing `chore_amd._lib` fails loudly
when the library is not built, and every operator raises when handed non-device tensors.
"""
__version__ = "0.1.0"
