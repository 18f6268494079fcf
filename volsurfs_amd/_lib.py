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


_CTYPES_OF = {"int": ctypes.c_int, "int32_t": ctypes.c_int32, "uint32_t": ctypes.c_uint32, "unsigned": ctypes.c_uint,
              "long long": ctypes.c_longlong, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
              "unsigned long long": ctypes.c_ulonglong, "size_t": ctypes.c_size_t, "float": ctypes.c_float,
              "double": ctypes.c_double, "uint8_t": ctypes.c_uint8}


def declared_prototypes():
    """{name: (restype, [argtypes])} of every function declared in include/volsurfs_hip.h, as ctypes types: every
    pointer is a c_void_p (tensors, structs by reference, host arrays), scalars by their C type.  The header is the one
    description of the boundary: with these set on the loaded library, ctypes converts a Python int to the width the
    callee reads (a small int handed to a `long long` parameter was a 32-bit argument before, wrapped by hand at each
    call site) and refuses a call with the wrong number or kind of arguments (VERDICT r5 missing #5)."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = {}
    for ret, name, params in re.findall(
            r"(?:^|\n)\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(vsa_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", src):
        ret = " ".join(ret.split())
        args = []
        for prm in params.split(","):
            prm = " ".join(prm.split())
            if prm in ("", "void"):
                continue
            if "*" in prm:
                args.append(ctypes.c_void_p)
                continue
            typ = prm.rsplit(" ", 1)[0] if " " in prm else prm
            if typ.startswith("const "):
                typ = typ[6:]
            if typ not in _CTYPES_OF:
                raise VolsurfsHipError(f"include/volsurfs_hip.h: {name}: no ctypes type for parameter '{prm}'")
            args.append(_CTYPES_OF[typ])
        if "*" in ret:
            res = ctypes.c_void_p
        elif ret in _CTYPES_OF:
            res = _CTYPES_OF[ret]
        else:
            raise VolsurfsHipError(f"include/volsurfs_hip.h: {name}: no ctypes type for return type '{ret}'")
        out[name] = (res, args)
    return out


# "0": no argtypes (rounds 1-5: every argument converted by _conv alone)
USE_ARGTYPES = os.environ.get("VSA_CTYPES_ARGTYPES", "1") != "0"


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
        protos = declared_prototypes() if USE_ARGTYPES else {}
        for name in declared_symbols():
            fn = getattr(_lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = ctypes.c_int
            if name in protos:
                fn.restype, fn.argtypes = protos[name]
    return _lib


def _conv(a, typed=False):
    """Convert a Python / torch argument to what ctypes takes.  typed: the function has argtypes — scalars go as they
    are (ctypes converts them to the declared width; a ctypes scalar object hands over its value, so that a call site
    that still wraps an `int` parameter's value in c_longlong, or the reverse, is converted instead of refused)."""
    import torch
    if a is None:
        return None if typed else ctypes.c_void_p(0)
    if isinstance(a, torch.Tensor):
        return ctypes.c_void_p(a.data_ptr())
    if typed:
        if isinstance(a, bool):
            return int(a)
        if isinstance(a, ctypes._SimpleCData) and not isinstance(a, (ctypes.c_void_p, ctypes.c_char_p, ctypes.c_wchar_p)):
            return a.value
        return a
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
    typed = fn.argtypes is not None
    if kernel_events is not None:
        import torch
        # keep the queue busy while the host prepares the launch: with an idle queue the start
        # event's timestamp is taken before the kernel has even been submitted, and the host's
        # launch latency would be counted as kernel time
        torch.cuda._sleep(400000)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*[_conv(x, typed) for x in args])
        b.record()
        kernel_events.setdefault(name, []).append((a, b))
    else:
        rc = fn(*[_conv(a, typed) for a in args])
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
