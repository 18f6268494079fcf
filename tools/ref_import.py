"""Import helper for the *reference* Python tree (this container only).

TEST INFRASTRUCTURE.  Used only by tools/make_golden.py to generate the small
golden fixtures committed under tests/golden/.  Nothing here ships to the GPU
box at run time and nothing from /root/reference is copied into this repo:
the reference modules are imported in place, with placeholder modules standing
in for third-party packages that are absent from this image (SURVEY.md
Appendix A).
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

REFERENCE_ROOT = "/root/reference"


class _Placeholder(types.ModuleType):
    """A module whose every attribute is another callable placeholder."""

    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []  # looks like a package so that sub-imports work

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        full = f"{self.__name__}.{item}"
        mod = sys.modules.get(full)
        if mod is None:
            mod = _Placeholder(full)
            sys.modules[full] = mod
        setattr(self, item, mod)
        return mod

    def __call__(self, *a, **k):
        return _Placeholder(self.__name__ + "()")

    def __mro_entries__(self, bases):  # allow `class X(placeholder)`
        return (object,)


ABSENT = [
    "volsurfs",
    "permutohedral_encoding",
    "tinycudann",
    "raytracelib",
    "apex",
    "apex.optimizers",
    "open3d",
    "open3d.visualization",
    "wandb",
    "hjson",
    "piq",
    "trimesh",
    "pymeshlab",
    "xatlas",
    "skimage",
    "skimage.measure",
    "cv2",
    "mvdatasets",
    "mvdatasets.utils",
    "mvdatasets.utils.raycasting",
    "mvdatasets.utils.tensor_mesh",
    "mvdatasets.utils.mesh",
    "mvdatasets.utils.images",
    "mvdatasets.utils.profiler",
    "mvdatasets.utils.tensor_reel",
    "mvdatasets.utils.tensor_texture",
    "mvdatasets.utils.virtual_cameras",
    "mvdatasets.geometry",
    "mvdatasets.geometry.contraction",
    "mvdatasets.geometry.primitives",
    "mvdatasets.geometry.primitives.bounding_sphere",
    "mvdatasets.geometry.primitives.bounding_box",
]


def install_placeholders(overrides=None):
    """Register placeholder modules; `overrides` maps a dotted module name to
    a dict of attributes that replace the placeholders (e.g. our own CPU
    restatement standing in for tinycudann)."""
    overrides = overrides or {}
    for name in ABSENT:
        if name not in sys.modules:
            sys.modules[name] = _Placeholder(name)
    for name, attrs in overrides.items():
        mod = sys.modules.get(name)
        if mod is None:
            mod = _Placeholder(name)
            sys.modules[name] = mod
        for k, v in attrs.items():
            setattr(mod, k, v)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
