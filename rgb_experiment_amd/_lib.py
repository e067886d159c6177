"""ctypes binding of librgbx_hip.so (C ABI: include/rgbx_hip.h).

There is no CPU fallback: if the library is missing, or a tensor is not on a HIP device, the
callers raise ``RuntimeError`` (the reference's sweep scripts treat ``RuntimeError`` as a failed
run, examples/all_dataset_baseline.py:65).
"""
import ctypes
import os

import torch  # must be imported first: its bundled libamdhip64.so.7 is the one HIP runtime in the process

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librgbx_hip.so")

_P = ctypes.c_void_p
_I64 = ctypes.c_int64
_I = ctypes.c_int
_F = ctypes.c_float

class RowSplit(ctypes.Structure):
    """rgbx_row_split_t"""
    _fields_ = [("threshold", ctypes.c_int32), ("n_chunks", ctypes.c_int32), ("n_long", ctypes.c_int32),
                ("chunk_begin", ctypes.c_void_p), ("chunk_end", ctypes.c_void_p), ("chunk_row", ctypes.c_void_p),
                ("long_row", ctypes.c_void_p),
                ("long_chunk_ptr", ctypes.c_void_p), ("partial", ctypes.c_void_p)]


class CeEpilogue(ctypes.Structure):
    """rgbx_ce_epilogue_t"""
    _fields_ = [("y", ctypes.c_void_p), ("mask", ctypes.c_void_p), ("grad_scale", ctypes.c_void_p),
                ("stats", ctypes.c_void_p), ("scratch", ctypes.c_void_p), ("mask_groups", ctypes.c_int32)]


class SpmmEpilogue(ctypes.Structure):
    """rgbx_spmm_epilogue_t"""
    _fields_ = [("ce", ctypes.c_void_p), ("n_classes", ctypes.c_int64), ("out_colsums", ctypes.c_void_p),
                ("stats_ws", ctypes.c_void_p), ("stats_ws_bytes", ctypes.c_size_t)]


class FusedLayer(ctypes.Structure):
    """rgbx_fused_layer_t"""
    _fields_ = [("rowptr", _P), ("col", _P), ("w", _P), ("rs", _P), ("x", _P), ("ldx", _I64),
                ("x_blk_cols", _I64), ("x_blk_stride", _I64), ("wt", _P), ("x_root", _P), ("ldr", _I64),
                ("xr_blk_cols", _I64), ("xr_blk_stride", _I64), ("wt_root", _P), ("bias", _P), ("out", _P),
                ("ldo", _I64), ("out_blk", _P), ("ob_cols", _I64), ("ob_stride", _I64), ("z_out", _P), ("ldz", _I64),
                ("pre_scale", _P), ("pre_shift", _P), ("pre_rowsum", _P), ("out_colsums", _P), ("stats_ws", _P),
                ("stats_ws_bytes", ctypes.c_size_t), ("ce", _P), ("N", _I64), ("K", _I64), ("Nout", _I64),
                ("split", _P), ("w_pos", _P), ("z_pos_out", _P)]


# name -> argtypes, exactly the declarations of include/rgbx_hip.h
SIGNATURES = {
    "rgbx_csr_workspace_bytes": [_I64, _I64, ctypes.POINTER(ctypes.c_size_t)],
    "rgbx_csr_build": [_P, _P, _I64, _I64, _I, _P, _P, _P, _P, ctypes.c_size_t, _P],
    "rgbx_deg_inv_sqrt_f32": [_P, _I64, _P, _P],
    "rgbx_gcn_norm_f32": [_P, _P, _I64, _P, _P, _P],
    "rgbx_inv_degree_f32": [_P, _I64, _P, _P],
    "rgbx_spmm_csr_f32": [_P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _I64, _I64, _I64, _F, _F, _P, _P],
    "rgbx_spmm_csr_epilogue_supported": [_I64],
    "rgbx_spmm_csr_epilogue_f32": [_P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _I64, _I64, _I64, _F, _F, _P, _P, _P],
    "rgbx_spmm_csr_short_rows_supported": [_I64],
    "rgbx_spmm_csr_short_rows_f32": [_P, _P, _P, _P, _I64, _P, _P, _I64, _I64, _I64, _P],
    "rgbx_spmm_linear_supported": [_I64, _I64, _I],
    "rgbx_spmm_linear_stats_workspace_bytes": [_I64, _I64, ctypes.POINTER(ctypes.c_size_t)],
    "rgbx_spmm_linear_f32": [_P, _P, _P, _P, _P, _I64, _P, _P, _I64, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _P, _P,
                             ctypes.c_size_t, _P, _I64, _I64, _I64, _P, _P],
    "rgbx_fused_layer_f32": [_P, _P],
    "rgbx_blocked_to_rows_f32": [_P, _I64, _I64, _P, _I64, _I64, _I64, _P, _P],
    "rgbx_appnp_f32": [_P, _P, _P, _P, _I64, _P, _P, _I64, _I64, _I64, _I, _F, _P, _P],
    "rgbx_dagnn_gate_fwd_f32": [_P, _I64, _P, _I64, _I64, _P, _P, _P, _I64, _I64, _I64, _I, _P],
    "rgbx_dagnn_gate_bwd_workspace_bytes": [_I64, ctypes.POINTER(ctypes.c_size_t)],
    "rgbx_dagnn_gate_bwd_f32": [_P, _I64, _P, _I64, _I64, _P, _P, _P, _I64, _P, _I64, _P, _I64, _I64, _P, _P, _P,
                                ctypes.c_size_t, _I64, _I64, _I, _P],
    "rgbx_gat_scores_f32": [_P, _I64, _P, _P, _P, _P, _I64, _I, _I, _P],
    "rgbx_gat_scores_bwd_scratch_floats": [_I64, _I, _I, ctypes.POINTER(ctypes.c_int64)],
    "rgbx_gat_scores_bwd_f32": [_P, _I64, _P, _P, _I64, _P, _P, _P, _I64, _P, _P, _P, _I64, _I64, _I, _I, _P],
    "rgbx_gat_aggregate_fwd_f32": [_P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _I64, _I, _I, _F, _P,
                                   _P],
    "rgbx_gat_edge_softmax_f32": [_P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I64, _P, _P],
    "rgbx_gat_bwd_dst_f32": [_P, _P, _P, _I64, _P, _P, _P, _P, _P, _I64, _P, _I64, _P, _P, _I64, _I, _I,
                             _F, _P],
    "rgbx_gat_bwd_prep_f32": [_P, _P, _P, _P, _I64, _P, _P, _I64, _P, _P, _P, _F, _P, _I64, _I, _I, _P],
    "rgbx_gat_bwd_src_f32": [_P, _P, _P, _I64, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _P, _I64, _I, _I, _F, _P, _P],
    "rgbx_gemm_tn_workspace_bytes": [_I64, _I64, _I64, ctypes.POINTER(ctypes.c_size_t)],
    "rgbx_gemm_tn_f32": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I64, _I64, _F, _P, ctypes.c_size_t, _P],
    "rgbx_bn_scratch_doubles": [_I64, _I64, ctypes.POINTER(ctypes.c_int64)],
    "rgbx_bn_stats_f32": [_P, _I64, _I64, _I64, _P, _P, _I64, _P],
    "rgbx_bn_finalize_f32": [_P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _I64, _P],
    "rgbx_fold_bn_linear_f32": [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _I64, _I64, _P],
    "rgbx_bn_bwd_finalize_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P],
    "rgbx_affine_cols_f32": [_P, _I64, _P, _P, _P, _I64, _I64, _I64, _P],
    "rgbx_bn_bwd_reduce_f32": [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _P, _P, _I64, _P],
    "rgbx_bn_bwd_apply_f32": [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P],
    "rgbx_masked_nll_scratch_doubles": [_I64, _I, ctypes.POINTER(ctypes.c_int64)],
    "rgbx_masked_nll_fwd_f32": [_P, _I64, _P, _P, _I64, _I64, _P, _P, _I64, _I, _P],
    "rgbx_masked_ce_fwd_f32": [_P, _I64, _P, _P, _I64, _I64, _P, _P, _I64, _P],
    "rgbx_masked_ce_fwd_blocked_f32": [_P, _I64, _I64, _P, _P, _P, _I64, _I64, _P, _P, _I64, _P],
    "rgbx_masked_ce_bwd_f32": [_P, _I64, _P, _P, _I64, _I64, _P, _P, _I64, _P],
    "rgbx_masked_nll_bwd_f32": [_P, _P, _I64, _I64, _P, _P, _I64, _P],
    "rgbx_coalesce_workspace_bytes": [_I64, _I64, _I, ctypes.POINTER(ctypes.c_size_t)],
    "rgbx_coalesce_keys_i64": [_P, _P, _I64, _I64, _I, _P, _P, _P, ctypes.c_size_t, _P],
    "rgbx_split_edge_keys_i64": [_P, _P, _I64, _I64, _P, _P, _P],
    "rgbx_gather_rows_f32": [_P, _I64, _P, _I64, _I64, _P, _I64, _P],
    "rgbx_scatter_add_rows_f32": [_P, _I64, _P, _I64, _I64, _P, _I64, _P],
    "rgbx_paced_copy_f32": [_P, _P, _I64, _I, _I, _P],
    "rgbx_py_random_shuffle_i64": [_I64, _I64, _P],
}
EXPORTS = ["rgbx_version", "rgbx_last_error_string"] + list(SIGNATURES)

_lib = None


def bind(path, strict=True):
    """ctypes handle of a build of the library with every entry point's signature declared. strict=False (kernel A/B
    tools binding an OLDER build): entry points that build does not have are skipped."""
    lib = ctypes.CDLL(path)
    lib.rgbx_version.restype = _I
    lib.rgbx_version.argtypes = []
    lib.rgbx_last_error_string.restype = ctypes.c_char_p
    lib.rgbx_last_error_string.argtypes = []
    for name, argtypes in SIGNATURES.items():
        if not strict and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.restype = _I
        fn.argtypes = argtypes
    return lib


def load():
    """Load the library once; raise RuntimeError (never fall back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"rgb_experiment_amd: {LIB_PATH} is not built — run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C rgb_experiment_amd/csrc`). There is no CPU fallback.")
    _lib = bind(LIB_PATH)
    return _lib


def use(lib):
    """Kernel A/B experiments (tools/ab_lib.py): route the calls through another build bound with `bind`."""
    global _lib
    _lib = lib


def check(rc, what):
    if rc != 0:
        msg = load().rgbx_last_error_string().decode(errors="replace")
        kind = "argument error" if rc < 0 else "hipError"
        raise RuntimeError(f"{what} failed ({kind} {rc}): {msg}")


def require_device(*tensors):
    """The product path runs on the GPU only."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "rgb_experiment_amd: the message-passing path runs in HIP kernels on an MI355X device; "
                f"got a tensor on '{t.device}'. There is no CPU fallback (oracle/ is test-only).")


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


def mat(t, name="tensor"):
    """Row-major fp32 matrix -> (pointer, leading dimension in elements)."""
    if t.dtype != torch.float32 or t.dim() != 2:
        raise RuntimeError(f"{name}: expected a 2-D float32 tensor, got {t.dtype} {tuple(t.shape)}")
    if t.size(1) > 1 and t.stride(1) != 1:
        raise RuntimeError(f"{name}: rows must be contiguous (stride {t.stride()})")
    ld = t.stride(0) if t.size(0) > 1 else max(t.size(1), 1)
    return t.data_ptr(), ld
