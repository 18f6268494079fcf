"""The C-ABI library builds, loads and exports every symbol include/*.h declares."""
import ctypes
import os
import subprocess

from volsurfs_amd import _lib


def test_library_builds_and_exports_all_declared_symbols():
    _lib.build()
    assert os.path.exists(_lib.LIB_PATH)
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    names = _lib.declared_symbols()
    assert "vsa_composite_dense_fwd" in names
    for n in names:
        assert hasattr(cdll, n), f"{n} declared in include/volsurfs_hip.h but not exported"
    assert cdll.vsa_version() >= 1


def test_exported_symbols_are_c_linkage():
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for n in _lib.declared_symbols():
        assert n in exported
