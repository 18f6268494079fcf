"""RCCL called directly (ctypes on librccl.so, the library torch.distributed's "nccl" backend itself loads): the
gradient all-reduces of the data-parallel step are enqueued ON the caller's stream — right behind the device-flag
wait that releases them (parallel.OverlappedStep) — instead of going through ProcessGroupNCCL's own stream and its
two cross-stream hand-offs per collective.  torch.distributed stays the control plane: it carries the 128-byte
unique id from rank 0 to the others when the communicator is formed.  (The reference is single-GPU)"""
import ctypes
import os

import torch

_lib = None

NCCL_FLOAT16, NCCL_FLOAT32, NCCL_BFLOAT16, NCCL_SUM = 6, 7, 9, 0
_DTYPES = {torch.float32: NCCL_FLOAT32, torch.float16: NCCL_FLOAT16, torch.bfloat16: NCCL_BFLOAT16}


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


class RcclError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if not os.path.exists(path):
            raise RcclError(f"{path} not found (a ROCm build of PyTorch ships it)")
        L = ctypes.CDLL(path)
        L.ncclGetErrorString.restype = ctypes.c_char_p
        L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        L.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RcclError(f"{what} failed: {lib().ncclGetErrorString(rc).decode()}")


class RcclComm:
    """One communicator over the ranks of a torch.distributed group (default: the world).  Every rank constructs it
    at the same point of the program (it is a collective: the unique id travels through the group)."""

    def __init__(self, rank, world, group=None):
        uid = _UniqueId()
        err = None
        # Every rank first shows that it can call the library at all and the ranks AGREE on that (MIN over the group)
        # before anyone enters the collective ncclCommInitRank: a rank that failed to load librccl.so would raise
        # locally while the others hang in the init for ever.
        try:
            lib()
            ok = 1
        except Exception as e:
            ok, err = 0, e
        if world > 1:
            import torch.distributed as dist
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 0:
                raise RcclError(f"librccl is not usable on every rank (this rank: {err or 'ok'}): no direct communicator")
        elif not ok:
            raise RcclError(str(err))
        err = None
        if rank == 0:
            try:
                _check(lib().ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
            except Exception as e:          # (the other ranks must still be told: they wait in the broadcast below)
                err = e
        if world > 1:
            import torch.distributed as dist
            box = [bytes(uid.internal) if (rank == 0 and err is None) else None]
            dist.broadcast_object_list(box, src=0, group=group)
            if box[0] is None:
                raise RcclError(f"rank 0 could not create an RCCL unique id ({err})")
            ctypes.memmove(ctypes.byref(uid), box[0], 128)
        elif err is not None:
            raise RcclError(str(err))
        self.comm = ctypes.c_void_p()
        _check(lib().ncclCommInitRank(ctypes.byref(self.comm), world, uid, rank), "ncclCommInitRank")
        self.rank, self.world = rank, world

    def all_reduce_sum_(self, t, stream=None):
        """In-place sum of the contiguous CUDA tensor `t` over the ranks, enqueued on `stream` (default: the current
        stream).  Asynchronous: ordered like any other launch on that stream."""
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in _DTYPES:
            raise RcclError(f"all_reduce_sum_: contiguous CUDA f32 / f16 / bf16 tensors only, got {t.dtype} {t.device}")
        s = stream if stream is not None else torch.cuda.current_stream()
        _check(lib().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _DTYPES[t.dtype], NCCL_SUM, self.comm,
                                   ctypes.c_void_p(s.cuda_stream)), "ncclAllReduce")

    def destroy(self):
        if getattr(self, "comm", None):
            lib().ncclCommDestroy(self.comm)
            self.comm = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
