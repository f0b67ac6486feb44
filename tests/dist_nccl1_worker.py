"""1-rank RCCL group: exercises Transport's device-pointer callbacks (pointer wrapping through
__cuda_array_interface__, ExternalStream ordering, RCCL all-reduce) exactly as the library calls
them.  Launched by tests/test_distributed.py; not collected by pytest."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from aspect_amd import parallel  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("WORLD_SIZE", "1")
    ranks = parallel.Ranks(backend="nccl", device=dev)
    if ranks.world == 1:   # Ranks only creates a group for world > 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29549")
        ranks.dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    tr = parallel.Transport(ranks)
    assert tr.device_buffers == 1
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        x = torch.arange(16, dtype=torch.float64, device=dev) * 0.5
        x += 1.0
    want = (torch.arange(16, dtype=torch.float64) * 0.5 + 1.0)
    rc = tr._allreduce(None, x.data_ptr(), 16, side.cuda_stream)
    side.synchronize()
    ok = rc == 0 and torch.equal(x.cpu(), want)
    # exchange with no messages is a no-op that must not fail
    rc = tr._exchange(None, 0, None, None, None, None, None, side.cuda_stream)
    ok = ok and rc == 0
    # the library's own RCCL transport on a 1-rank communicator: id hand-over, ncclCommInitRank,
    # ncclAllReduce on the library stream inside every dot product of a whole solve
    import numpy as np
    from aspect_amd import capi, cases, partition
    g = cases.nsinker_box(2, n_sinkers=4, ratio=1e4).build()
    loc = partition.LocalProblem(g, 0, 1)
    plain = capi.Solver(capi.load(), g)
    native = capi.Solver(capi.load(), loc, transport=parallel.RcclTransport(ranks))
    xa, ra, _ = plain.solve()
    xb, rb, _ = native.solve()
    same = ra.n_cheap_iterations == rb.n_cheap_iterations and \
        np.linalg.norm(xa - xb) <= 1e-6 * np.linalg.norm(xa)
    print(f"[nccl1] native RCCL transport: iterations {rb.n_cheap_iterations} vs {ra.n_cheap_iterations}", flush=True)
    ok = ok and same
    # the two coarsest levels marked as replicated: their dot products stay local, and the restriction onto the last of
    # them ends in an ncclAllReduce of the whole coarse defect on the library stream (agmg_ctx::allreduce_vec)
    loc2 = partition.LocalProblem(g, 0, 1)
    for lev in loc2.hier.levels[:2]:
        lev.replicated = True
    rep = capi.Solver(capi.load(), loc2, transport=parallel.RcclTransport(ranks))
    xc, rc_, _ = rep.solve()
    same2 = ra.n_cheap_iterations == rc_.n_cheap_iterations and np.linalg.norm(xa - xc) <= 1e-6 * np.linalg.norm(xa)
    print(f"[nccl1] native RCCL transport, replicated coarse levels: iterations {rc_.n_cheap_iterations} vs {ra.n_cheap_iterations}", flush=True)
    ok = ok and same2
    rep.close()
    plain.close()
    native.close()
    print(f"[nccl1] {'ok' if ok else 'FAIL'}", flush=True)
    ranks.dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
