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
    assert lib.rgbx_version() == 100


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
