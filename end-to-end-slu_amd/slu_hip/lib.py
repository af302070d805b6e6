"""ctypes binding of libslu_hip.so (C ABI declared in include/slu_hip.h).

The library is built in-tree by csrc/build.sh (hipcc --offload-arch=gfx950).  There is no CPU or
PyTorch fallback: if the shared object is missing, or a call fails, this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLU_HIP_LIB: another build of the same ABI (the probe build of tools/gru_probe.py); the product never sets it
LIB_PATH = os.environ.get("SLU_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libslu_hip.so")

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_f32 = ctypes.c_float
c_f64 = ctypes.c_double
c_u64 = ctypes.c_uint64
c_sz = ctypes.c_size_t
vp = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/slu_hip.h one to one
SIGNATURES = {
    "slu_version": (c_int, []),
    "slu_last_error": (ctypes.c_char_p, []),
    "slu_device_check": (c_int, []),
    "slu_device_arch": (ctypes.c_char_p, []),
    "slu_stream_create_cu_range": (c_int, [c_i64, c_i64, vp]),
    "slu_store_u64": (c_int, [vp, vp, c_i64, vp]),
    "slu_stage_inputs": (c_int, [vp, vp, vp, vp, vp, c_i64, vp, c_i64, vp]),
    "slu_multi_max": (c_int, []),
    "slu_copy_multi": (c_int, [vp, vp, vp, c_i64, vp]),
    "slu_scale_multi": (c_int, [vp, vp, c_i64, vp, vp]),
    "slu_absmax_multi": (c_int, [vp, vp, c_i64, vp, vp]),
    "slu_pcm16_to_f32": (c_int, [vp, vp, c_i64, c_f32, vp]),
    "slu_pool_act_fwd": (c_int, [vp, vp, vp, c_i64, c_i64, c_i64, c_i64, c_int, c_f32, c_i64, c_i64, vp]),
    "slu_pool_act_bwd": (c_int, [vp, vp, vp, vp, c_i64, c_i64, c_i64, c_i64, c_f32, c_i64, c_i64, vp]),
    "slu_gru_cell_fwd": (c_int, [vp, vp, vp, c_i64, vp, c_i64, vp, vp, vp, c_f32, c_u64, c_u64, vp, c_u64, c_i64, c_i64, vp]),
    "slu_gru_cell_bwd": (c_int, [vp, c_i64, vp, vp, vp, c_i64, vp, vp, vp, c_i64, vp, c_f32, c_u64, c_u64, vp, c_u64,
                                 c_i64, c_i64, vp]),
    "slu_attention_fwd": (c_int, [vp, c_i64, c_i64, vp, c_i64, c_i64, vp, c_i64, vp, c_i64, vp, c_f32, c_i64, c_i64, c_i64,
                                  c_i64, vp]),
    "slu_attention_bwd": (c_int, [vp, c_i64, c_i64, vp, c_i64, c_i64, vp, c_i64, vp, c_i64, vp, vp, vp, vp, c_i64, c_f32,
                                  c_i64, c_i64, c_i64, c_i64, vp]),
    "slu_logsoftmax_dot_fwd": (c_int, [vp, vp, c_i64, vp, vp, c_i64, c_i64, vp]),
    "slu_logsoftmax_dot_bwd": (c_int, [vp, vp, c_i64, vp, vp, c_i64, vp, c_i64, c_i64, vp]),
    "slu_neg_mean_f32": (c_int, [vp, vp, c_i64, vp]),
    "slu_fill_scaled_f32": (c_int, [vp, c_i64, vp, c_f32, vp]),
    "slu_broadcast_rows_f32": (c_int, [vp, vp, c_i64, c_i64, c_i64, vp]),
    "slu_sinc_filters_fwd": (c_int, [vp, vp, vp, c_i64, c_i64, c_f64, vp]),
    "slu_sinc_filters_bwd": (c_int, [vp, vp, vp, vp, vp, c_i64, c_i64, c_f64, vp]),
    "slu_wconv_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "slu_wconv_fwd": (c_int, [vp, vp, vp, vp, vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int,
                              c_int, c_f32, c_i64, c_i64, vp, c_sz, vp]),
    "slu_wconv_bwd_act": (c_int, [vp, vp, vp, vp, c_i64, c_i64, c_i64, c_int, c_int, c_f32, c_i64,
                                  c_i64, vp]),
    "slu_wconv_bwd_data": (c_int, [vp, vp, vp, c_i64, c_i64, c_i64, c_i64, c_i64, vp, c_sz, vp]),
    "slu_wconv_bwd_weight_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64, c_i64, c_i64]),
    "slu_wconv_bwd_weight": (c_int, [vp, vp, vp, vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, vp,
                                     c_sz, vp]),
    "slu_gemm_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "slu_gemm_f32": (c_int, [vp, c_i64, c_i64, vp, c_i64, c_i64, vp, c_i64, c_i64, vp, c_i64, c_i64,
                             c_i64, c_int, vp, c_sz, vp]),
    "slu_split_bf16": (c_int, [vp, c_i64, vp, c_i64, c_i64, c_i64, c_int, vp]),
    "slu_gemm_bf16_pack_bytes": (c_sz, [c_i64, c_i64, c_int]),
    "slu_gemm_bf16_pack": (c_int, [vp, c_i64, c_i64, vp, c_i64, c_i64, c_int, vp]),
    "slu_gemm_bf16": (c_int, [vp, c_i64, c_i64, vp, vp, vp, c_i64, c_i64, c_i64, c_i64, c_int, vp]),
    "slu_gemm_bf16_a32": (c_int, [vp, c_i64, vp, vp, vp, c_i64, c_i64, c_i64, c_i64, c_int, vp]),
    "slu_gemm_tn_bf16_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "slu_gemm_tn_bf16": (c_int, [vp, c_i64, vp, c_i64, vp, c_i64, c_i64, c_i64, c_i64, c_int, vp, c_sz, vp]),
    "slu_wconv_bf16_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_int]),
    "slu_wconv_fwd_bf16": (c_int, [vp, vp, c_i64, vp, vp, vp, vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_f32,
                                   c_i64, c_i64, vp, c_i64, vp, c_sz, c_int, c_int, vp, c_int, c_f32, vp]),
    "slu_gru_seq_fwd_bf16": (c_int, [vp, vp, vp, vp, vp, vp, vp, vp, c_i64, c_i64, vp, vp, c_i64, c_i64, c_i64, c_i64, c_int,
                                     c_int, vp]),
    "slu_gru_seq_fwd_pool_bf16": (c_int, [vp, vp, vp, vp, vp, vp, vp, c_i64, vp, c_f32, vp, c_i64, c_i64, vp, vp, c_i64,
                                          c_i64, c_i64, c_i64, c_int, c_int, vp]),
    "slu_dropout_bits": (c_int, [vp, c_f32, c_u64, c_u64, vp, c_i64, c_u64, c_i64, c_i64, c_i64, vp]),
    "slu_comm_version": (c_int, []),
    "slu_comm_unique_id": (c_int, [vp]),
    "slu_comm_init": (c_int, [vp, vp, c_i64, c_i64]),
    "slu_comm_allreduce_f32": (c_int, [vp, vp, c_i64, vp]),
    "slu_comm_allreduce_f64": (c_int, [vp, vp, c_i64, vp]),
    "slu_comm_destroy": (c_int, [vp]),
    "slu_comm_allreduce_group": (c_int, [vp, vp, c_i64, vp, c_i64, vp]),
    "slu_comm_ipc_window_bytes": (c_i64, [c_i64]),
    "slu_comm_ipc_window_create": (c_int, [c_i64, c_i64, vp, vp]),
    "slu_comm_ipc_window_open": (c_int, [vp, vp]),
    "slu_comm_ipc_window_close": (c_int, [vp]),
    "slu_comm_ipc_window_destroy": (c_int, [vp]),
    "slu_comm_allreduce_ipc": (c_int, [vp, c_i64, c_i64, c_i64, vp, c_i64, vp, c_i64, vp]),
    "slu_comm_ipc_status": (c_int, [vp, vp]),
    "slu_comm_ipc_set_spin_limit": (c_int, [vp, c_i64]),
    "slu_comm_ipc_resident_workgroups": (c_int, [c_i64, vp, vp]),
    "slu_comm_ipc_max_wait": (c_int, [vp, vp]),
    "slu_comm_ipc_window_touch": (c_int, [vp, c_i64, c_i64, c_i64, vp]),
    "slu_gru_proj_supported": (c_int, [c_i64, c_i64, c_i64, c_i64, c_i64]),
    "slu_gru_proj_state_words": (c_i64, [c_i64, c_i64]),
    "slu_gru_proj_seq_fwd": (c_int, [vp, c_i64, vp, c_i64, vp, vp, vp, vp, vp, vp, vp, vp, c_i64, c_i64, c_i64, c_i64, c_i64, vp, c_i64, vp]),
    "slu_gemm_tn_batched": (c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, c_i64, vp, c_i64, c_i64, vp, vp]),
    "slu_gemm_tn_splitk_workspace_bytes": (c_sz, [vp, vp, vp, c_i64]),
    "slu_gemm_tn_batched_splitk": (c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, c_i64, vp, c_i64, c_i64, vp, vp, c_sz, vp, c_i64, vp]),
    "slu_gemm_tn_splitk_workspace_bytes_wg": (c_sz, [vp, vp, vp, c_i64, c_i64]),
    "slu_gemm_tn_batched_splitk_wg": (c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, c_i64, vp, c_i64, c_i64, vp, vp, c_sz, vp, c_i64, c_i64, vp]),
    "slu_gemm_small_batched": (c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, c_i64, vp]),
    "slu_colsum_f32": (c_int, [vp, c_i64, vp, c_i64, c_i64, c_int, vp]),
    "slu_gru_reserve_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64]),
    "slu_gru_bias_tiles": (c_i64, [c_i64, c_i64, c_i64, c_i64]),
    "slu_frame_ce_fwd": (c_int, [vp, vp, c_i64, c_i64, c_i64, c_int, vp, vp, vp]),
    "slu_adam_max_tensors": (c_int, []),
    "slu_adam_multi": (c_int, [vp, vp, vp, vp, vp, c_i64, c_int, vp, c_f64, c_f64, c_f64, c_f64, c_f64, vp, vp]),
    "slu_adam_advance_step": (c_int, [vp, c_u64, vp]),
    "slu_gru_seq_fwd": (c_int, [vp, vp, vp, vp, vp, vp, vp, c_i64, c_i64, c_i64, c_i64, vp]),
    "slu_gru_seq_bwd": (c_int, [vp, vp, vp, vp, vp, vp, vp, c_i64, c_i64, c_i64, c_i64, vp]),
    "slu_dropout_pool_fwd": (c_int, [vp, vp, c_i64, c_i64, vp, c_f32, c_u64, c_u64, vp, c_i64, c_u64, c_int,
                                     c_i64, vp, c_i64, c_i64, c_i64, vp]),
    "slu_dropout_pool_fwd_planes": (c_int, [vp, vp, c_i64, c_i64, vp, c_f32, c_u64, c_u64, vp, c_i64, c_u64, c_int,
                                            c_i64, vp, c_i64, c_int, c_i64, c_i64, c_i64, vp]),
    "slu_dropout_pool_bwd": (c_int, [vp, vp, vp, vp, c_i64, c_i64, c_f32, c_u64, c_u64, vp, c_i64, c_u64,
                                     c_int, c_i64, vp, c_i64, c_i64, c_i64, vp]),
    "slu_cls_maxpool_ce_fwd": (c_int, [vp, vp, vp, vp, ctypes.POINTER(c_i64), c_i64, vp, vp, vp, vp, vp, vp, vp, vp,
                                       c_f32, c_u64, c_u64, vp, vp, c_i64, c_i64, c_i64, vp]),
    "slu_cls_maxpool_ce_bwd": (c_int, [vp, vp, vp, vp, vp, vp, vp, vp, c_f32, c_u64, c_u64, vp, c_i64, c_i64, c_i64, c_i64,
                                       vp]),
}

_lib = None
ABI_VERSION = 9          # SLU_ABI_VERSION of include/slu_hip.h


class SluHipError(RuntimeError):
    pass


def load():
    """Loads libslu_hip.so (once).  Raises SluHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (libamdhip64); it must be the one already resident when
    # libslu_hip.so is mapped, or the process ends up with two runtimes and no visible device.
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
        raise SluHipError(
            "libslu_hip.so not found at %s — build it with end-to-end-slu_amd/csrc/build.sh "
            "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the ABI and this table disagree
        fn.restype = res
        fn.argtypes = args
    if lib.slu_version() != ABI_VERSION:
        raise SluHipError("libslu_hip.so ABI version %d, expected %d (rebuild with csrc/build.sh)"
                          % (lib.slu_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().slu_last_error().decode("utf-8", "replace")
        raise SluHipError("%s failed (status %d): %s" % (what or "libslu_hip call", rc, msg))


def require_gfx950():
    lib = load()
    check(lib.slu_device_check(), "slu_device_check")
