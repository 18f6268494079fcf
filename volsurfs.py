"""`import volsurfs` — the name the reference's Python imports its native extension under
(/root/reference/src/PyBridge.cxx:19 `PYBIND11_MODULE(volsurfs, m)`), resolved to the
MI355X mirror of that operator surface:

    from volsurfs import VolumeRendering, RaySampler          # utils/background.py:3,5
    from volsurfs import OccupancyGrid, RaySamplesPacked      # utils/sampling.py:3, methods/nerf.py:10-11
    import volsurfs; os.path.dirname(volsurfs.__file__)       # params/cmd_params.py:2,8-9

This file is a single-file module (not a package directory) on purpose: the reference locates
its repository root as the directory holding the extension module, with `config/` next to it —
copy or symlink this file and `volsurfs_amd/` into the root of a reference checkout and those
imports resolve without editing the reference.  All compute is in
volsurfs_amd/libvolsurfs_hip.so; importing fails loudly if it is not built.
"""
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
if _here not in _sys.path:
    _sys.path.insert(0, _here)

from volsurfs_amd import _lib as _lib  # noqa: E402
from volsurfs_amd.volsurfs import (  # noqa: E402,F401
    OccupancyGrid, RaySampler, RaySamplesPacked, VolumeRendering)

__all__ = ["OccupancyGrid", "RaySampler", "RaySamplesPacked", "VolumeRendering"]

_lib.lib()      # dlopen now: `import volsurfs` must not succeed without the HIP library
