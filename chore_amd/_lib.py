"""ctypes binding of libchore_hip.so (include/chore_hip.h).

This is the only place Python touches the native library.  There is deliberately NO fallback:
if the shared object is missing the import raises, and `check()` turns every non-zero return code
into a RuntimeError carrying chore_last_error().
"""
import ctypes
import os

import torch  # noqa: F401  -- FIRST: PyTorch ships its own libamdhip64; loading libchore_hip.so before it would bind the
#                              library to the system ROCm's copy and the process would hold two HIP runtimes (the second one
#                              finds no device).  With torch loaded the dlopen below resolves to the runtime torch uses.
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_longlong, c_size_t, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHORE_HIP_LIB: another build of the same library (A/B runs of two kernel versions on one GPU box)
LIB_PATH = os.environ.get("CHORE_HIP_LIB") or os.path.join(_HERE, "csrc", "libchore_hip.so")

F32, BF16, F16X3, F16 = 0, 1, 2, 3
HEADS_X3 = 0x100      # OR into a query dtype: heads on the fp16 matrix cores with split operands (include/chore_hip.h)


class WeightDesc(ctypes.Structure):
    _fields_ = [("name", c_char_p), ("ptr", c_void_p), ("numel", c_int64)]


class EncoderCfg(ctypes.Structure):
    _fields_ = [("in_channels", c_int), ("num_stack", c_int), ("num_hourglass", c_int),
                ("hourglass_dim", c_int)]


# every exported symbol of include/chore_hip.h with its signature (restype, argtypes)
SIGNATURES = {
    "chore_version": (c_int, []),
    "chore_create": (c_int, [POINTER(c_void_p), c_int]),
    "chore_destroy": (c_int, [c_void_p]),
    "chore_last_error": (c_char_p, [c_void_p]),
    "chore_cu_count": (c_int, [c_void_p]),
    "chore_stream_create_cu_mask": (c_int, [c_void_p, POINTER(ctypes.c_uint32), c_int, POINTER(c_void_p)]),
    "chore_stream_destroy": (c_int, [c_void_p, c_void_p]),
    "chore_heads_arena_bytes": (c_size_t, [c_int]),
    "chore_heads_pack": (c_int, [c_void_p, POINTER(WeightDesc), c_int, c_int, c_void_p, c_void_p]),
    "chore_query_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int,
                                c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                c_void_p, POINTER(c_float), c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "chore_query_fwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "chore_query_fwd_ws": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int,
                                   c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                   c_void_p, POINTER(c_float), c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "chore_sample_features": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                      POINTER(c_float), c_void_p, c_void_p, c_void_p, c_void_p]),
    "chore_query_bwd_points": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int,
                                       c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                       c_void_p, POINTER(c_float), c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "chore_query_train_bytes": (c_size_t, [c_int, c_int]),
    "chore_query_fwd_train": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                      c_void_p, POINTER(c_float), c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p]),
    "chore_query_bwd_train": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                      c_void_p, POINTER(c_float), c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "chore_scatter_features": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                       POINTER(c_float), c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "chore_encoder_arena_bytes": (c_size_t, [POINTER(EncoderCfg), c_int]),
    "chore_encoder_pack": (c_int, [c_void_p, POINTER(EncoderCfg), POINTER(WeightDesc), c_int, c_int,
                                   c_void_p, c_void_p]),
    "chore_encoder_workspace_bytes": (c_size_t, [POINTER(EncoderCfg), c_int, c_int, c_int, c_int]),
    "chore_encode_fwd": (c_int, [c_void_p, POINTER(EncoderCfg), c_void_p, c_int, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_size_t, POINTER(c_void_p), c_int, c_void_p,
                                 c_void_p, c_void_p]),
    "chore_smpl_arena_bytes": (c_size_t, [c_int, c_int, c_int]),
    "chore_smpl_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "chore_smpl_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                POINTER(c_int), c_void_p, c_void_p]),
    "chore_smpl_lbs_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chore_smpl_lbs_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_int, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chore_landmarks_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "chore_landmarks_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "chore_so3_aux_bytes": (c_size_t, [c_int]),
    "chore_so3_project_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "chore_so3_project_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "chore_conv2d_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "chore_gn_stats_bytes": (c_size_t, [c_int]),
    "chore_gn_stats": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "chore_gn_relu_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                  c_void_p]),
    "chore_conv2d_fwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chore_amax_bytes": (c_size_t, []),
    "chore_absmax_f32": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "chore_conv2d_bwd_data": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "chore_conv2d_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "chore_conv2d_bwd_weight": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chore_convblock_saved_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "chore_convblock_out_stats_offset": (c_size_t, [c_int]),
    "chore_convblock_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "chore_convblock_grad_floats": (c_size_t, [c_int, c_int]),
    "chore_convblock_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chore_convblock_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chore_debug_nan_counts": (c_int, [c_void_p]),
    "chore_fit_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float,
                                    c_float, c_float, c_void_p, c_void_p]),
    "chore_fit_adam_step_acc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                        c_void_p, c_float, c_float, c_float, c_float, c_void_p, c_int, c_void_p]),
    "chore_fit_weighted_sum_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "chore_fit_weighted_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "chore_fit_weighted_sum_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chore_fit_smpl_terms_fwd": (c_int, [c_void_p] * 11 + [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "chore_fit_smpl_terms_bwd": (c_int, [c_void_p] * 11 + [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chore_fit_point_terms_workspace_bytes": (c_size_t, [c_int, c_int]),
    "chore_fit_point_terms_fwd": (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p]),
    "chore_fit_point_terms_bwd": (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p]),
    "chore_fit_obj_transform_fwd": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p, c_void_p]),
    "chore_fit_obj_transform_bwd": (c_int, [c_void_p] * 6 + [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chore_fit_obj_terms_workspace_bytes": (c_size_t, [c_int]),
    "chore_fit_obj_terms_fwd": (c_int, [c_void_p] * 5 + [c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chore_fit_obj_terms_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    "chore_fit_rot_noise": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_longlong, c_void_p, c_void_p]),
    "chore_fit_stop_rule": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "chore_train_loss_workspace_bytes": (c_size_t, []),
    "chore_train_loss": (c_int, [c_void_p] * 11 + [c_int, c_int, c_float, c_void_p, c_float] + [c_void_p] * 5 + [c_int, c_void_p,
                                                                                                               c_void_p]),
    "chore_upadd_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "chore_up2_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "chore_avgpool2_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "chore_avgpool2_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "chore_stem_workspace_bytes": (c_size_t, [c_int]),
    "chore_stem_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p]),
    "chore_stem_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "chore_stem_bwd_weight": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    "chore_heads_wgrad_floats": (c_size_t, []),
    "chore_heads_wgrad_workspace_bytes": (c_size_t, []),
    "chore_heads_wgrad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_void_p]),
    "chore_gemm_tn_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "chore_gemm_tn_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p]),
    "chore_gn_relu_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "chore_gn_relu_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "chore_eval_chamfer_workspace_bytes": (c_size_t, [c_int, c_int]),
    "chore_eval_chamfer": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "chore_eval_procrustes": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "chore_eval_apply_similarity": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "chore_gen_clamp_mask": (c_int, [c_void_p, c_void_p, c_int, c_float, c_int, c_int, c_void_p, c_void_p]),
    "chore_gen_surface_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_int, c_int, c_void_p, c_void_p]),
    "chore_gen_surface_step_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                             c_int, c_void_p, POINTER(c_float), c_int, c_float, c_void_p, c_void_p]),
    "chore_gen_compact": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "chore_gen_append": (c_int, [c_void_p, c_void_p, c_longlong, c_longlong, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_longlong, c_longlong, c_longlong, c_int, c_int, c_int, c_int, c_void_p]),
    "chore_gen_advance": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "chore_gen_resample": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                   c_int, c_float, c_void_p, c_void_p]),
    "chore_prep_masks2bbox": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "chore_prep_resize_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "chore_prep_crop_compose": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_void_p, c_void_p]),
    "chore_prep_crop_compose_mean": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_double, ctypes.c_double,
                                             ctypes.c_double, ctypes.c_double, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "chore_collision_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "chore_collision_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "chore_collision_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "chore_sil_project_fwd": (c_int, [c_void_p] * 9 + [c_int, c_void_p, c_float, c_float, c_int, c_int, c_int, c_void_p, c_void_p]),
    "chore_sil_project_bwd": (c_int, [c_void_p] * 8 + [c_int, c_void_p, c_float, c_float, c_int, c_int, c_int] + [c_void_p] * 7),
    "chore_silhouette_workspace_bytes": (c_size_t, [c_int, c_int]),
    "chore_silhouette_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    "chore_silhouette_bwd": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "chore_contact_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "chore_contact_fwd": (c_int, [c_void_p] * 7 + [c_int] * 4 + [c_float, c_void_p, c_void_p, c_void_p]),
    "chore_contact_bwd": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p] * 5),
    "chore_profile_enable": (c_int, [c_void_p, c_int]),
    "chore_profile_read": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(ctypes.c_double),
                                   POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int64)]),
}


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -m chore_amd.build` "
            "(hipcc --offload-arch=gfx950). chore_amd has no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load_library()


class ChoreError(RuntimeError):
    pass


_handles = {}


def handle(device_index: int) -> c_void_p:
    """one chore_handle per (process, device) -- and per host THREAD: a handle carries an error string, lazily set kernel
    attributes and (training) a side stream with its events, none of which two threads should share; the pipelined fit
    (recon_fit_behave.fit_recon(pipeline=True)) drives a second stream from a second thread"""
    import threading
    if threading.current_thread() is not threading.main_thread():
        device_index = (device_index, threading.get_ident())
    h = _handles.get(device_index)
    if h is None:
        h = c_void_p()
        rc = lib.chore_create(ctypes.byref(h), device_index if isinstance(device_index, int) else device_index[0])
        if rc != 0:
            raise ChoreError(f"chore_create(device={device_index}) failed with {rc}: "
                             f"{lib.chore_last_error(None).decode()}")
        _handles[device_index] = h
    return h


def check(rc: int, h: c_void_p, what: str):
    if rc != 0:
        msg = lib.chore_last_error(h)
        raise ChoreError(f"{what} failed with {rc}: {msg.decode() if msg else ''}")


def cu_masked_stream(device_index: int, n_cus: int):
    """a torch stream whose kernels run on n_cus of the device's compute units, the same number in every XCD
    (chore_stream_create_cu_mask; mask bit i = CU i / 8 of XCD i % 8).  The stream lives as long as the process."""
    import torch
    h = handle(device_index)
    total = lib.chore_cu_count(h)
    if not 0 < n_cus <= total:
        raise ChoreError(f"cu_masked_stream: {n_cus} of {total} compute units")
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(n_cus):
        mask[i // 32] |= 1 << (i % 32)
    st = c_void_p()
    check(lib.chore_stream_create_cu_mask(h, mask, words, ctypes.byref(st)), h, "chore_stream_create_cu_mask")
    return torch.cuda.ExternalStream(st.value, device=torch.device("cuda", device_index))


def profile_enable(device_index: int, on: bool):
    h = handle(device_index)
    check(lib.chore_profile_enable(h, 1 if on else 0), h, "chore_profile_enable")


def profile_read(device_index: int):
    """-> {class: dict(ms, flops, bytes, launches)} accumulated since profile_enable"""
    h = handle(device_index)
    n = 48
    names = (c_char_p * n)()
    ms, fl, by = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)()
    la = (c_int64 * n)()
    k = lib.chore_profile_read(h, n, names, ms, fl, by, la)
    return {names[i].decode(): dict(ms=ms[i], flops=fl[i], bytes=by[i], launches=la[i]) for i in range(k)}


def make_descs(named_tensors):
    """[(name, fp32 contiguous device tensor)] -> (ctypes array, keepalive list)"""
    arr = (WeightDesc * len(named_tensors))()
    keep = []
    for i, (name, t) in enumerate(named_tensors):
        bname = name.encode()
        keep.append((bname, t))
        arr[i].name = bname
        arr[i].ptr = t.data_ptr()
        arr[i].numel = t.numel()
    return arr, keep
