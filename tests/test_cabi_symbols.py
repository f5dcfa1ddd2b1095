"""The C-ABI library loads on a CPU-only box and exports every symbol include/srlz.h declares (no compute calls)."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "srlz.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srlz_[a-zA-Z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert len(syms) >= 40 and "srlz_conv64_fwd" in syms and "srlz_adam_step" in syms


def test_library_exports_every_declared_symbol(cabi):
    lib = ctypes.CDLL(cabi.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, "libsrlz_hip.so does not export: %s" % missing


def test_binding_covers_header(cabi):
    missing = [s for s in declared_symbols() if s not in cabi.EXPORTED]
    assert not missing, "srlz/_cabi.py has no prototype for: %s" % missing


def test_version_and_error_text(cabi):
    assert cabi.version() == cabi.ABI_VERSION == 104
    d = cabi.Conv64Desc(1, 8, 8, 9, 9, 3, 1, 1, 0)  # inconsistent output size -> host-side rejection
    assert cabi._lib.srlz_conv64_fwd_tiles(ctypes.byref(d)) == -1
    assert "inconsistent" in cabi.error_text()
