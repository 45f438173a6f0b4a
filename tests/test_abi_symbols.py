"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/ssgpu.h declares (no compute calls here)."""
import ctypes
import os
import re

from supersonic_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ssgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssgpu_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libssgpu.so does not export %s" % n
    bound = {n for n, _r, _a in _lib.SYMBOLS}
    assert set(names) == bound, (set(names) ^ bound)


def test_abi_version_and_bind_only_context():
    lib = _lib.load()
    assert lib.ssgpu_abi_version() == 10
    h = ctypes.c_void_p()
    assert lib.ssgpu_ctx_create(-1, ctypes.byref(h)) == 0
    assert lib.ssgpu_ctx_synchronize(h) == _lib.ERROR_NO_DEVICE
    lib.ssgpu_ctx_destroy(h)
