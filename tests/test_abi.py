"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/nif_hip.h declares; the ctypes structs mirror the header; no compute is called."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "nif_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nif_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_header_symbol():
    from nif_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail("libnif_hip.so missing: run `python __graft_entry__.py` (hipcc cross-compiles gfx950 without a GPU)")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _header_functions()
    assert len(names) >= 30
    for nm in names:
        assert hasattr(lib, nm), "header declares %s but the library does not export it" % nm
    # and the ctypes table binds exactly the header's functions
    assert sorted(_lib.SIGNATURES) == names


def test_abi_version_and_struct_sizes():
    from nif_amd import _lib
    lib = _lib.load()
    assert lib.nif_abi_version() == _lib.NIF_ABI_VERSION
    assert ctypes.sizeof(_lib.nif_cfg) == 4 * 16 + 4 * 8
    assert ctypes.sizeof(_lib.nif_tensor_desc) == 48 + 8 + 4 + 4
    assert ctypes.sizeof(_lib.nif_adam) == 16


def test_no_cpu_fallback_without_device():
    """Without a HIP device the product must fail loudly, not compute on the host."""
    from nif_amd import _lib
    lib = _lib.load()
    if lib.nif_device_count() > 0:
        pytest.skip("a GPU is visible here")
    import nif_amd
    from tests.cfgs import cfg_ms
    _, cs, cp = cfg_ms()
    m = nif_amd.NIFMultiScale(cs, cp)  # construction needs no GPU (like the reference needs no data)
    with pytest.raises(_lib.NifError):
        m.build().predict([[0.0, 0.0]])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "nif_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, fn)).read()
                assert "oracle" not in src.replace("no oracle", ""), "%s mentions the oracle" % fn
