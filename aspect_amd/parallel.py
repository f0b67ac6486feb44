"""Rank plumbing for bench.py / multi-rank runs: one process per GPU, torch.distributed
(backend "nccl" = RCCL on ROCm; "gloo" on CPU for tests).  The data path has no collective yet
(every rank owns an independent subdomain-sized problem); what crosses ranks is timing only."""
from __future__ import annotations

import os


class Ranks:
    def __init__(self, backend=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.backend = backend
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            # a rank that fails before a collective must end the job, not leave the others waiting forever
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=int(os.environ.get("AGMG_DIST_TIMEOUT", "600")))
            dist.init_process_group(backend or "gloo", rank=self.rank, world_size=self.world, **kw)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        if self.device is not None and self.device.type == "cuda":
            self.torch.cuda.synchronize()

    def max_over_ranks(self, x: float) -> float:
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device if self.backend == "nccl" else None)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x: float) -> float:
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device if self.backend == "nccl" else None)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_concat(self, a):
        """float64 arrays of equal shape on every rank -> their concatenation in rank order (set-up paths)."""
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.float64)
        if self.world == 1:
            return a.copy()
        dev = self.device if self.backend == "nccl" else None
        mine = self.torch.from_numpy(a).to(dev) if dev is not None else self.torch.from_numpy(a)
        parts = [self.torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        return np.concatenate([t.cpu().numpy() for t in parts], axis=0)

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def aggregate_throughput(units_per_rank_step: float, steps: int, seconds_max: float, world: int) -> float:
    """whole-job units/s: units all ranks processed / max-over-ranks time (weak scaling)."""
    return world * units_per_rank_step * steps / seconds_max


# ---------------------------------------------------------------------------------------------
# Transport behind agmg_set_communicator (include/agmg.h): the library asks for two things only,
# a neighbour exchange (ghost update / compress of one operator application) and a sum all-reduce
# of a few scalars (dot products).  Here both go through torch.distributed: backend "nccl" (RCCL
# over xGMI) on device pointers, ordered on the library's own HIP stream with no host
# synchronisation; backend "gloo" on the host buffers the library stages when device_buffers = 0.
# ---------------------------------------------------------------------------------------------
import ctypes as C

import numpy as np

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32),
                          C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                          C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)


class CommStruct(C.Structure):
    _fields_ = [("rank", C.c_int32), ("size", C.c_int32), ("device_buffers", C.c_int32),
                ("user", C.c_void_p), ("exchange", EXCHANGE_FN), ("allreduce_sum", ALLREDUCE_FN)]


class _DevView:
    """Minimal __cuda_array_interface__ carrier: lets torch wrap a raw device pointer."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False),
                                         "version": 2, "strides": None}


def exchange_host(dist, torch, msgs):
    """msgs: list of (peer, send ndarray | None, recv ndarray | None); non-blocking pairs, then wait."""
    ops, keep = [], []
    for peer, snd, rcv in msgs:
        if snd is not None and snd.size:
            t = torch.from_numpy(snd)
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, peer))
        if rcv is not None and rcv.size:
            t = torch.from_numpy(rcv)
            keep.append(t)
            ops.append(dist.P2POp(dist.irecv, t, peer))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


class Transport:
    """Owns the C callbacks (keeps them alive) for one Ranks group."""

    def __init__(self, ranks: Ranks, device_buffers=None):
        self.r = ranks
        self.torch, self.dist = ranks.torch, ranks.dist
        # gloo with device_buffers = 1 (tests on one GPU): this class stages the device buffers
        # through host tensors itself, exercising the library's device-pointer protocol
        self.device_buffers = (1 if ranks.backend == "nccl" else 0) if device_buffers is None else int(device_buffers)
        self.on_device_wire = ranks.backend == "nccl"
        self._views = {}
        self._streams = {}
        self.n_exchange = self.n_allreduce = 0
        self._ex = EXCHANGE_FN(self._exchange)
        self._ar = ALLREDUCE_FN(self._allreduce)
        self.struct = CommStruct(ranks.rank, ranks.world, self.device_buffers, None, self._ex, self._ar)

    # -- helpers ------------------------------------------------------------------------------
    def _dev(self, ptr, n):
        key = (ptr, n)
        t = self._views.get(key)
        if t is None:
            t = self.torch.as_tensor(_DevView(ptr, n), device=self.r.device)
            self._views[key] = t
        return t

    def _stream(self, handle):
        s = self._streams.get(handle)
        if s is None:
            s = self.torch.cuda.ExternalStream(handle, device=self.r.device)
            self._streams[handle] = s
        return s

    # -- callbacks ----------------------------------------------------------------------------
    def _exchange(self, user, n, peer, send, n_send, recv, n_recv, stream):
        try:
            self.n_exchange += 1
            if self.device_buffers:
                dist = self.dist
                with self.torch.cuda.stream(self._stream(stream)):
                    if self.on_device_wire:
                        ops = []
                        for i in range(n):
                            if n_send[i] > 0:
                                ops.append(dist.P2POp(dist.isend, self._dev(send[i], n_send[i]), peer[i]))
                            if n_recv[i] > 0:
                                ops.append(dist.P2POp(dist.irecv, self._dev(recv[i], n_recv[i]), peer[i]))
                        if ops:
                            for req in dist.batch_isend_irecv(ops):
                                req.wait()      # stream-ordered: makes `stream` wait for the transfers
                    else:
                        msgs, back = [], []
                        for i in range(n):
                            snd = self._dev(send[i], n_send[i]).cpu().numpy() if n_send[i] > 0 else None
                            rcv = np.zeros(n_recv[i]) if n_recv[i] > 0 else None
                            msgs.append((peer[i], snd, rcv))
                            if rcv is not None:
                                back.append((self._dev(recv[i], n_recv[i]), rcv))
                        exchange_host(self.dist, self.torch, msgs)
                        for t, rcv in back:
                            t.copy_(self.torch.from_numpy(rcv))
                        self.torch.cuda.current_stream().synchronize()
            else:
                msgs = []
                for i in range(n):
                    snd = np.ctypeslib.as_array(C.cast(send[i], C.POINTER(C.c_double)), (n_send[i],)) if n_send[i] > 0 else None
                    rcv = np.ctypeslib.as_array(C.cast(recv[i], C.POINTER(C.c_double)), (n_recv[i],)) if n_recv[i] > 0 else None
                    msgs.append((peer[i], snd, rcv))
                exchange_host(self.dist, self.torch, msgs)
            return 0
        except Exception as e:  # never let an exception cross the C ABI
            import sys
            print(f"[aspect_amd transport] exchange failed on rank {self.r.rank}: {e!r}", file=sys.stderr, flush=True)
            return -1

    def _allreduce(self, user, values, n, stream):
        try:
            self.n_allreduce += 1
            if self.device_buffers:
                with self.torch.cuda.stream(self._stream(stream)):
                    t = self._dev(values, n)
                    if self.on_device_wire:
                        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
                    else:
                        h = t.cpu()
                        self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
                        t.copy_(h)
                        self.torch.cuda.current_stream().synchronize()
            else:
                a = np.ctypeslib.as_array(C.cast(values, C.POINTER(C.c_double)), (n,))
                self.dist.all_reduce(self.torch.from_numpy(a), op=self.dist.ReduceOp.SUM)
            return 0
        except Exception as e:
            import sys
            print(f"[aspect_amd transport] all-reduce failed on rank {self.r.rank}: {e!r}", file=sys.stderr, flush=True)
            return -1


class RcclTransport:
    """Marker transport for capi.Solver: the library's own RCCL transport
    (agmg_set_communicator_rccl).  torch.distributed only carries the 128-byte unique id from rank 0
    to the other ranks; every exchange and all-reduce then runs inside the library on its stream."""
    native_rccl = True

    def __init__(self, ranks: Ranks):
        self.r = ranks
        self.struct = CommStruct(ranks.rank, ranks.world, 1, None, EXCHANGE_FN(), ALLREDUCE_FN())
        self.n_exchange = self.n_allreduce = -1   # not observable from the host

    def unique_id(self, lib):
        torch, dist = self.r.torch, self.r.dist
        buf = (C.c_uint8 * 128)()
        if self.r.rank == 0:
            rc = lib.rccl_get_unique_id(buf)
            if rc:
                raise RuntimeError(lib.last_error().decode())
        t = torch.tensor(list(buf), dtype=torch.uint8)
        if self.r.world > 1:
            if self.r.backend == "nccl":
                t = t.to(self.r.device)
            dist.broadcast(t, src=0)
        return [int(v) for v in t.cpu().tolist()]
