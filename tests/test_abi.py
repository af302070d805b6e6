"""CPU: the C-ABI shared library loads and exports every symbol include/slu_hip.h declares; the
argument-validation paths that need no GPU behave as documented."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "slu_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slu_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_table_agree():
    from slu_hip import lib
    assert header_functions() == sorted(lib.SIGNATURES)


def test_library_loads_and_exports_every_symbol():
    from slu_hip import lib
    L = lib.load()
    for name in header_functions():
        assert hasattr(L, name), name
    assert L.slu_version() == lib.ABI_VERSION == 9
    assert isinstance(L.slu_last_error(), bytes)


def test_size_queries_and_argument_validation_without_gpu():
    from slu_hip import lib
    L = lib.load()
    # reserve: D*T*ceil(B/16)*(H/16)*5*256 floats
    assert L.slu_gru_reserve_bytes(300, 64, 128, 2) == 2 * 300 * 4 * 8 * 5 * 256 * 4
    assert L.slu_gru_reserve_bytes(7, 3, 16, 2) == 2 * 7 * 1 * 1 * 5 * 256 * 4
    assert L.slu_wconv_workspace_bytes(80, 1, 401) >= 101 * 5 * 64 * 4
    assert L.slu_gemm_workspace_bytes(19200, 768, 256) == 0          # enough tiles: no split-K
    assert L.slu_gemm_workspace_bytes(384, 128, 19200) > 0
    # null pointers / bad sizes are rejected before anything touches the device
    rc = L.slu_gru_seq_fwd(None, None, None, None, None, None, None, 10, 4, 128, 2, None)
    assert rc == -1 and b"null" in L.slu_last_error()
    rc = L.slu_sinc_filters_fwd(1, 1, 1, 80, 400, 16000.0, None)      # even filter length
    assert rc == -1 and b"odd" in L.slu_last_error()
    with pytest.raises(lib.SluHipError):
        lib.check(rc, "slu_sinc_filters_fwd")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from slu_hip import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.SluHipError, match="no CPU fallback"):
        lib.load()


def test_graft_entry_build_checks_the_current_abi_version():
    """__graft_entry__.build() is the driver's "does it build" check: it must accept whatever SLU_ABI_VERSION the header
    carries (a hard-coded number there went stale once), and the library on disk must be the one the binding expects."""
    from slu_hip import lib
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "lib.ABI_VERSION" in src and not re.search(r"ABI_VERSION\s*==\s*\d", src)
    header = open(os.path.join(ROOT, "include", "slu_hip.h")).read()
    assert int(re.search(r"#define\s+SLU_ABI_VERSION\s+(\d+)", header).group(1)) == lib.ABI_VERSION
