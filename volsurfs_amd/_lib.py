"""ctypes loader for libvolsurfs_hip.so — the C-ABI declared in include/volsurfs_hip.h.

The product path has NO CPU fallback: if the shared library is missing or a
call returns non-zero, a RuntimeError is raised (the reference only prints
kernel errors, src/VolumeRendering.cu:64-75; SURVEY §8b "Errors" asks the
replacement never to continue silently).
"""
import ctypes
import os
import re
import subprocess

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG_DIR)
LIB_PATH = os.path.join(_PKG_DIR, "libvolsurfs_hip.so")
HEADER_PATH = os.path.join(_ROOT, "include", "volsurfs_hip.h")
CSRC_DIR = os.path.join(_PKG_DIR, "csrc")

_lib = None


class VolsurfsHipError(RuntimeError):
    pass


def build(jobs=4, verbose=False):
    """Compile every HIP/C++ source for gfx950 into the in-tree .so (hipcc
    cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, f"-j{jobs}"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise VolsurfsHipError("building libvolsurfs_hip.so failed")
    return LIB_PATH


def declared_symbols():
    """Names of every function declared in include/volsurfs_hip.h."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vsa_[a-z0-9_]+)\s*\(", src)))


def lib():
    """The loaded library (ctypes.CDLL).  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VolsurfsHipError(
                f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C volsurfs_amd/csrc`. There is no CPU fallback.")
        # torch ships its own libamdhip64; import it first so that this library
        # binds to the SAME HIP runtime instance (two runtimes in one process do
        # not share devices / streams).
        import torch  # noqa: F401
        _lib = ctypes.CDLL(LIB_PATH)
        for name in declared_symbols():
            fn = getattr(_lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = ctypes.c_int
    return _lib


def _conv(a):
    """Convert a Python / torch argument to a ctypes value."""
    import torch
    if a is None:
        return ctypes.c_void_p(0)
    if isinstance(a, torch.Tensor):
        return ctypes.c_void_p(a.data_ptr())
    if isinstance(a, bool):
        return ctypes.c_int(int(a))
    if isinstance(a, int):
        return ctypes.c_longlong(a) if abs(a) > 0x7FFFFFFF else ctypes.c_int(a)
    if isinstance(a, float):
        return ctypes.c_float(a)
    return a  # already a ctypes object


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


kernel_events = None   # bench.py: {} -> every call is bracketed by events on the current stream


def kernel_ms():
    """Median duration per C-ABI entry point over the calls recorded in `kernel_events` (the median: one call
    that sat behind another stream's work — a communicator coming up, a collective of the previous step —
    would otherwise own the mean of three)."""
    import statistics
    import torch
    torch.cuda.synchronize()
    return {k: statistics.median(a.elapsed_time(b) for a, b in v) for k, v in (kernel_events or {}).items()}


def kernel_totals():
    """{entry point: (total ms, calls)} over the calls recorded in `kernel_events`."""
    import torch
    torch.cuda.synchronize()
    return {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in (kernel_events or {}).items()}


def call(name, *args):
    """Call a C-ABI entry point; raise on a non-zero status."""
    fn = getattr(lib(), name)
    if kernel_events is not None:
        import torch
        # keep the queue busy while the host prepares the launch: with an idle queue the start
        # event's timestamp is taken before the kernel has even been submitted, and the host's
        # launch latency would be counted as kernel time
        torch.cuda._sleep(400000)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*[_conv(x) for x in args])
        b.record()
        kernel_events.setdefault(name, []).append((a, b))
    else:
        rc = fn(*[_conv(a) for a in args])
    if rc != 0:
        raise VolsurfsHipError(f"{name} failed with status {rc}")
    return rc


def check_f32(t, *shape):
    import torch
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise VolsurfsHipError(
            f"expected a contiguous CUDA float32 tensor, got {t.dtype} {t.device} "
            f"contiguous={t.is_contiguous()}")
    if shape and tuple(t.shape) != tuple(shape):
        raise VolsurfsHipError(f"expected shape {shape}, got {tuple(t.shape)}")
    return t
