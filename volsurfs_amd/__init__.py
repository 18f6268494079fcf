"""volsurfs_amd — MI355X-native (gfx950) K-shell layered-mesh render hot path.

Host-side mirror of the reference's operator surface for the hot path named in
BASELINE.json; all compute is in libvolsurfs_hip.so (hand-written
HIP, C-ABI in include/volsurfs_hip.h).  There is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
