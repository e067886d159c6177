"""The C-ABI library builds, loads and exports every symbol include/rgbx_hip.h declares.
No compute is launched here (no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rgbx_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rgbx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    for must in ("rgbx_csr_build", "rgbx_spmm_csr_f32", "rgbx_appnp_f32", "rgbx_gat_aggregate_fwd_f32",
                 "rgbx_gat_bwd_dst_f32", "rgbx_gat_bwd_src_f32", "rgbx_gather_rows_f32", "rgbx_version",
                 "rgbx_last_error_string"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from rgb_experiment_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in rgbx_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == declared_symbols()
    assert lib.rgbx_version() == 501


def test_argument_errors_do_not_need_a_gpu():
    """Negative return codes come from host-side validation, before any launch."""
    from rgb_experiment_amd import _lib
    lib = _lib.load()
    rc = lib.rgbx_spmm_csr_f32(None, None, None, None, None, 4, None, 0, None, None, 4, 8, 4, 1.0, 0.0, None, None)
    assert rc == -1
    assert b"null" in lib.rgbx_last_error_string()
    n = ctypes.c_size_t(0)
    assert lib.rgbx_csr_workspace_bytes(-1, 4, ctypes.byref(n)) == -1
    assert lib.rgbx_csr_workspace_bytes(2**31, 4, ctypes.byref(n)) == -2
    assert lib.rgbx_csr_workspace_bytes(1000, 10, ctypes.byref(n)) == 0 and n.value >= 3 * 1010 * 4
    with pytest.raises(RuntimeError, match="argument error"):
        _lib.check(-1, "probe")


def test_shape_and_alignment_errors_of_the_fused_and_dense_entry_points():
    """Still host-side only: every rejected call returns before a launch (fake, never dereferenced pointers)."""
    from rgb_experiment_amd import _lib
    lib = _lib.load()
    p = 0x10000  # 16-byte aligned, non-null
    ok = lib.rgbx_spmm_linear_supported
    assert ok(128, 128, 0) and ok(128, 128, 1) and ok(4, 32, 0) and ok(256, 512, 0) and ok(64, 256, 1)
    assert not ok(130, 128, 0) and not ok(260, 128, 0) and not ok(128, 100, 0) and not ok(64, 512, 1) and not ok(0, 32, 0)
    call = lambda K, n_out, x=p, ldx=None, root=None: lib.rgbx_spmm_linear_f32(
        p, p, None, None, x, K if ldx is None else ldx, p, root, K, root, None, p, n_out, None, K, None, None, None, None,
        None, 0, None, 10, K, n_out, None, None)
    assert call(5, 32) == -5 and b"K" in lib.rgbx_last_error_string()          # RGBX_E_SHAPE
    assert call(128, 100) == -5
    assert call(64, 512, root=p) == -5                                           # root term: Nout <= 256
    assert call(128, 128, x=p + 4) == -3                                         # RGBX_E_ALIGN
    assert call(128, 128, ldx=64) == -1                                          # leading dimension < K
    assert lib.rgbx_spmm_linear_f32(p, p, None, None, p, 128, p, p, 128, None, None, p, 128, None, 128, None, None, None,
                                    None, None, 0, None, 10, 128, 128, None, None) == -1   # x_root without wt_root
    assert lib.rgbx_spmm_linear_f32(p, p, None, None, p, 128, p, None, 128, None, None, p, 128, None, 128, p, None, None,
                                    None, None, 0, None, 10, 128, 128, None, None) == -1   # pre_scale without shift / rowsum
    assert lib.rgbx_spmm_linear_f32(p, p, None, None, p, 128, p, None, 128, None, None, p, 128, None, 128, None, None, None,
                                    p, p, 16, None, 10, 128, 128, None, None) == -4        # statistics workspace too small
    ce = _lib.CeEpilogue(p, None, None, p, p)
    assert lib.rgbx_spmm_linear_f32(p, p, None, None, p, 128, p, None, 128, None, None, None, 256, None, 128, None, None,
                                    None, None, None, 0, ctypes.byref(ce), 10, 128, 256, None, None) == -5  # Nout > 128
    assert lib.rgbx_spmm_linear_f32(p, p, None, None, p, 128, p, None, 128, None, None, None, 128, None, 128, None, None,
                                    None, None, None, 0, ctypes.byref(_lib.CeEpilogue(None, None, None, p, p)), 10, 128,
                                    128, None, None) == -1                                   # epilogue without labels
    # the struct form (partitioned runs): dense mode, blocked layouts
    def layer(**kw):
        L = _lib.FusedLayer()
        base = dict(rowptr=None, x=p, ldx=128, wt=p, out=p, ldo=128, N=10, K=128, Nout=128)
        base.update(kw)
        for k, v in base.items():
            setattr(L, k, v)
        return lib.rgbx_fused_layer_f32(ctypes.byref(L), None)
    assert lib.rgbx_fused_layer_f32(None, None) == -1
    assert layer(x=None) == -1
    assert layer(out=None) == -1                                                  # nothing to write
    assert layer(x_blk_cols=48, x_blk_stride=480) == -1                           # 48 does not divide K = 128
    assert layer(x_blk_cols=32, x_blk_stride=322) == -1                           # stride % 4
    assert layer(rowptr=p, col=p, x_blk_cols=32, x_blk_stride=320) == -1          # blocked x only in dense mode
    assert layer(out=None, out_blk=p, ob_cols=48, ob_stride=480) == -1            # 48 does not divide Nout
    assert layer(out_blk=p, ob_cols=32, ob_stride=320, ce=ctypes.addressof(ce)) == -1   # loss epilogue: no blocked output
    assert layer(K=130) == -5
    # an aggregating launch with per-slot weights + loss epilogue (or a second aggregate) has no blocked-root form
    agg = dict(rowptr=p, col=p, w=p, x_root=p, wt_root=p, xr_blk_cols=32, xr_blk_stride=320)
    assert layer(ce=ctypes.addressof(ce), out=None, **agg) == -1 and b"blocked root" in lib.rgbx_last_error_string()
    assert layer(w_pos=p, z_pos_out=p, z_out=p, ldz=128, **agg) == -1
    # new in 4.0.0: edge-list ingest
    n = ctypes.c_size_t(0)
    assert lib.rgbx_coalesce_workspace_bytes(-1, 4, 0, ctypes.byref(n)) == -1
    assert lib.rgbx_coalesce_workspace_bytes(10, 2**31, 0, ctypes.byref(n)) == -2
    rc = lib.rgbx_coalesce_workspace_bytes(1000, 100, 1, ctypes.byref(n))  # rocPRIM's size query of `unique` asks the
    assert (rc == 0 and n.value >= 2 * 2000 * 8) or rc > 0                  # device for its configuration: a hipError here
    assert lib.rgbx_coalesce_keys_i64(None, None, 5, 10, 0, p, p, p, 1 << 20, None) == -1
    assert lib.rgbx_coalesce_keys_i64(p, p, 5, 10, 0, p, None, p, 1 << 20, None) == -1    # no counts
    assert lib.rgbx_split_edge_keys_i64(None, p, 5, 10, p, p, None) == -1
    assert lib.rgbx_split_edge_keys_i64(p, p, 0, 10, p, p, None) == 0                      # nothing to do
    assert lib.rgbx_blocked_to_rows_f32(p, 32, 320, p, 128, 10, 100, None, None) == -1  # 32 does not divide d = 100
    assert lib.rgbx_blocked_to_rows_f32(p, 32, 320, p, 64, 10, 128, None, None) == -1   # ldd < d
    assert lib.rgbx_blocked_to_rows_f32(None, 32, 320, p, 128, 10, 128, None, None) == -1
    assert lib.rgbx_blocked_to_rows_f32(p, 32, 320, p, 128, 10, 128, p + 4, None) == -1  # bias not 16-byte aligned
    assert lib.rgbx_blocked_to_rows_f32(p, 32, 320, p, 128, 0, 128, None, None) == 0    # nothing to do
    # weight-gradient GEMM: workspace too small / leading dimension
    n = ctypes.c_size_t(0)
    assert lib.rgbx_gemm_tn_workspace_bytes(1000, 128, 128, ctypes.byref(n)) == 0 and n.value >= 128 * 128 * 4
    assert lib.rgbx_gemm_tn_f32(p, 128, p, 128, p, 128, None, 1000, 128, 128, 1.0, p, 16, None) == -4   # RGBX_E_WS
    assert lib.rgbx_gemm_tn_f32(p, 64, p, 128, p, 128, None, 1000, 128, 128, 1.0, p, n.value, None) == -1
    # DAGNN hop mix: width, alignment, workspace
    assert lib.rgbx_dagnn_gate_fwd_f32(p, 6, p, 600, 6, p, None, p, 6, 100, 6, 2, None) == -5     # d % 4
    assert lib.rgbx_dagnn_gate_fwd_f32(p, 260, p, 26000, 260, p, None, p, 260, 100, 260, 2, None) == -5  # d > 256
    assert lib.rgbx_dagnn_gate_fwd_f32(p + 4, 8, p, 800, 8, p, None, p, 8, 100, 8, 2, None) == -3
    assert lib.rgbx_dagnn_gate_fwd_f32(p, 8, None, 800, 8, p, None, p, 8, 100, 8, 2, None) == -1  # K > 0 without hops
    assert lib.rgbx_dagnn_gate_fwd_f32(p, 8, p, 800, 8, p, None, None, 8, 100, 8, 2, None) == -1
    assert lib.rgbx_dagnn_gate_bwd_workspace_bytes(128, ctypes.byref(n)) == 0 and n.value >= 2048 * 129 * 4
    assert lib.rgbx_dagnn_gate_bwd_f32(p, 8, p, 800, 8, p, None, p, 8, p, 8, p, 800, 8, p, None, p, 16, 100, 8, 2,
                                       None) == -4
    # GAT: a head wider than 64 lanes x 4 floats
    assert lib.rgbx_gat_scores_f32(p, 1000, p, p, p, p, 10, 1, 1000, None) == -5


def test_product_refuses_cpu_tensors():
    import torch
    from rgb_experiment_amd.nn import GCNConv
    conv = GCNConv(4, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback|No CPU fallback|HIP"):
        conv(torch.randn(5, 4), torch.tensor([[0, 1], [1, 2]]))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rgb_experiment_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
