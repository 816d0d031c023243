"""Native RCCL communicator (csrc/comm/rccl_comm.cpp) behind the same flat-buffer facade as
`ddp.FlatDataParallel`.

`torch.distributed`'s "nccl" backend already is RCCL on ROCm and stays the default; this path talks to
librccl directly from our own C++ so collectives are enqueued on whatever HIP stream the trainer is
using (torch's current stream: they are captured by hipGraph capture like any kernel), with no
c10d work objects in between.  Bootstrap needs only a way to ship 128 bytes from rank 0 to the
others: the torch.distributed store when a process group exists, else a file path.

    comm = RcclCommunicator.from_env(device)        # RANK / WORLD_SIZE / MASTER_* as set by torchrun
    comm.all_reduce_sum_(net.G); opt.step(comm.grad_scale)
"""
from __future__ import annotations

import ctypes as C
import os
import time
from pathlib import Path
from typing import Optional

import torch

_LIB = Path(__file__).resolve().parent.parent / "_C" / "libddl_comm.so"
_DT = {torch.float32: 0, torch.bfloat16: 1, torch.int32: 2, torch.int64: 3, torch.uint8: 4}
SUM, MAX, MIN, AVG = 0, 1, 2, 3
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not _LIB.exists():
            raise ImportError(f"{_LIB} is missing: run `python build.py --only comm`")
        _lib = C.CDLL(str(_LIB))
        vp, ci, u64 = C.c_void_p, C.c_int, C.c_uint64
        _lib.ddl_comm_last_error.restype = C.c_char_p
        _lib.ddl_comm_unique_id.argtypes = [vp]
        _lib.ddl_comm_init.restype = vp
        _lib.ddl_comm_init.argtypes = [vp, ci, ci, ci]
        _lib.ddl_comm_all_reduce.argtypes = [vp, vp, u64, ci, ci, vp]
        _lib.ddl_comm_broadcast.argtypes = [vp, vp, u64, ci, ci, vp]
        _lib.ddl_comm_reduce_scatter.argtypes = [vp, vp, vp, u64, ci, ci, vp]
        _lib.ddl_comm_all_gather.argtypes = [vp, vp, vp, u64, ci, vp]
        _lib.ddl_comm_send.argtypes = [vp, vp, u64, ci, ci, vp]
        _lib.ddl_comm_recv.argtypes = [vp, vp, u64, ci, ci, vp]
        _lib.ddl_comm_barrier.argtypes = [vp, vp]
        _lib.ddl_comm_async_error.argtypes = [vp]
        _lib.ddl_comm_abort.argtypes = [vp]
        _lib.ddl_comm_destroy.argtypes = [vp]
    return _lib


class RcclError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise RcclError(_load().ddl_comm_last_error().decode("utf8", "replace"))


class RcclCommunicator:
    def __init__(self, unique_id: bytes, world: int, rank: int, device: torch.device,
                 comm_dtype: Optional[torch.dtype] = None):
        lib = _load()
        self.device = torch.device(device)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        buf = C.create_string_buffer(unique_id, len(unique_id))
        self._h = lib.ddl_comm_init(buf, world, rank, idx)
        if not self._h:
            raise RcclError(lib.ddl_comm_last_error().decode("utf8", "replace"))
        self.world, self.rank, self.comm_dtype = world, rank, comm_dtype
        self._staging = {}

    # ---- bootstrap ------------------------------------------------------------------------
    @staticmethod
    def new_unique_id() -> bytes:
        lib = _load()
        n = lib.ddl_comm_unique_id_bytes()
        buf = C.create_string_buffer(n)
        _chk(lib.ddl_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_env(cls, device, comm_dtype=None, id_file: Optional[str] = None, timeout: float = 120.0):
        """Rank / world from the torchrun environment; the unique id travels through the
        torch.distributed store if a process group is up, else through `id_file`."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        import torch.distributed as dist
        if world == 1:
            uid = cls.new_unique_id()
        elif dist.is_available() and dist.is_initialized():
            box = [cls.new_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
        else:
            if not id_file:
                raise RcclError("need an initialised torch.distributed group or id_file= to bootstrap")
            if rank == 0:
                uid = cls.new_unique_id()
                tmp = id_file + ".tmp"
                with open(tmp, "wb") as f:
                    f.write(uid)
                os.replace(tmp, id_file)
            else:
                t0 = time.time()
                while not os.path.exists(id_file):
                    if time.time() - t0 > timeout:
                        raise RcclError(f"timed out waiting for {id_file}")
                    time.sleep(0.05)
                uid = open(id_file, "rb").read()
        return cls(uid, world, rank, device, comm_dtype)

    # ---- helpers ---------------------------------------------------------------------------
    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    @staticmethod
    def _check(t: torch.Tensor):
        if not (t.is_cuda and t.is_contiguous() and t.dtype in _DT):
            raise ValueError(f"RCCL collectives need a contiguous cuda tensor of a supported dtype, got {t.dtype} "
                             f"{t.device} contiguous={t.is_contiguous()}")

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    # ---- the FlatDataParallel facade ------------------------------------------------------------
    def all_reduce_sum_(self, t: torch.Tensor, async_op: bool = False, op: int = SUM):
        self._check(t)
        if self.comm_dtype is not None and t.dtype != self.comm_dtype:
            st = self._staging.get((t.data_ptr(), t.numel()))
            if st is None:
                st = self._staging[(t.data_ptr(), t.numel())] = torch.empty_like(t, dtype=self.comm_dtype)
            st.copy_(t)
            _chk(_load().ddl_comm_all_reduce(self._h, st.data_ptr(), st.numel(), _DT[st.dtype], op, self._stream()))
            t.copy_(st)
            return None
        _chk(_load().ddl_comm_all_reduce(self._h, t.data_ptr(), t.numel(), _DT[t.dtype], op, self._stream()))
        return None

    def all_reduce_mean_(self, t: torch.Tensor):
        self.all_reduce_sum_(t, op=AVG)

    def broadcast_(self, *tensors, src: int = 0):
        for t in tensors:
            self._check(t)
            _chk(_load().ddl_comm_broadcast(self._h, t.data_ptr(), t.numel(), _DT[t.dtype], src, self._stream()))

    def all_gather_scalars(self, values: torch.Tensor) -> torch.Tensor:
        self._check(values)
        out = torch.empty((self.world,) + tuple(values.shape), dtype=values.dtype, device=values.device)
        _chk(_load().ddl_comm_all_gather(self._h, values.data_ptr(), out.data_ptr(), values.numel(),
                                         _DT[values.dtype], self._stream()))
        return out

    def reduce_scatter_sum(self, t: torch.Tensor) -> torch.Tensor:
        self._check(t)
        if t.numel() % self.world:
            raise ValueError("reduce_scatter needs numel divisible by world")
        n = t.numel() // self.world
        out = torch.empty(n, dtype=t.dtype, device=t.device)
        _chk(_load().ddl_comm_reduce_scatter(self._h, t.data_ptr(), out.data_ptr(), n, _DT[t.dtype], SUM,
                                             self._stream()))
        return out

    def all_gather_flat(self, shard: torch.Tensor, out: torch.Tensor):
        self._check(shard)
        self._check(out)
        if out.numel() != shard.numel() * self.world:
            raise ValueError("all_gather_flat: out must hold world shards")
        _chk(_load().ddl_comm_all_gather(self._h, shard.data_ptr(), out.data_ptr(), shard.numel(), _DT[shard.dtype],
                                         self._stream()))
        return out

    def send(self, t: torch.Tensor, dst: int):
        self._check(t)
        _chk(_load().ddl_comm_send(self._h, t.data_ptr(), t.numel(), _DT[t.dtype], dst, self._stream()))

    def recv(self, t: torch.Tensor, src: int):
        self._check(t)
        _chk(_load().ddl_comm_recv(self._h, t.data_ptr(), t.numel(), _DT[t.dtype], src, self._stream()))

    def barrier(self):
        _chk(_load().ddl_comm_barrier(self._h, self._stream()))
        torch.cuda.current_stream().synchronize()

    # ---- failure detection -----------------------------------------------------------------------
    def healthy(self) -> bool:
        """False once RCCL has recorded an asynchronous error (peer death, link failure)."""
        return _load().ddl_comm_async_error(self._h) == 0

    def abort(self):
        if self._h:
            _load().ddl_comm_abort(self._h)

    def close(self):
        if self._h:
            _load().ddl_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
