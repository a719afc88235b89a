"""CPU: the C-ABI shared library loads and exports every symbol include/lipvq.h declares, and the
ctypes table (lipvq-vae_amd/_capi.py) covers exactly that set.  No compute calls (no GPU here)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "lipvq.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lipvq_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = _declared()
    for must in ("lipvq_mlp3_f32", "lipvq_nearest_f32", "lipvq_lipschitz_scale_f32", "lipvq_mse_pair_f32",
                 "lipvq_mlp3_bwd_f32", "lipvq_wgrad_f32", "lipvq_scatter_add_f32", "lipvq_lipschitz_bwd_f32"):
        assert must in names


def test_library_exports_every_declared_symbol():
    so = ROOT / "lipvq-vae_amd" / "_lipvq_hip.so"
    assert so.exists(), "build the HIP library first (__graft_entry__.build())"
    import torch  # noqa: F401  (maps torch's libamdhip64 first, as the product does)
    lib = ctypes.CDLL(str(so))
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/lipvq.h but not exported"
    lib.lipvq_abi_version.restype = ctypes.c_int
    assert lib.lipvq_abi_version() == 1


def test_ctypes_table_matches_header():
    import lipvq_vae_amd
    from lipvq_vae_amd import _capi
    assert sorted(_capi.SIGNATURES) == _declared()
    # pure host-side entry points are callable without a GPU
    assert _capi.lib.lipvq_mlp3_packed_floats(7, 64, 128, 64) == (2 * 4 * 64 + 64) + (4 * 32 * 64 + 128) + (2 * 64 * 64 + 64)
    assert _capi.lib.lipvq_mse_workspace_bytes() > 0
    # one partial slab per chunk of rows: 64-row chunks below 16 384 rows, 1024-row chunks from 262 144 rows on
    assert _capi.lib.lipvq_wgrad_workspace_bytes(5000, 128, 64) == 79 * (128 * 64 + 128) * 4
    assert _capi.lib.lipvq_wgrad_workspace_bytes(300000, 128, 64) == 293 * (128 * 64 + 128) * 4


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """The product must not fall back to anything when the extension is absent."""
    import importlib.util
    src = (ROOT / "lipvq-vae_amd" / "_capi.py").read_text()
    (tmp_path / "_capi.py").write_text(src)
    spec = importlib.util.spec_from_file_location("capi_copy", tmp_path / "_capi.py")
    mod = importlib.util.module_from_spec(spec)
    with pytest.raises(ImportError, match="no CPU fallback"):
        spec.loader.exec_module(mod)
