"""CPU worker (gloo, launched by tests/test_distributed.py; not collected by pytest): each rank
applies the serial CPU oracle to its local sub-mesh, then the interface partial sums are compressed
to the owners and the owner values sent back to the ghosts through aspect_amd.parallel's host
transport, following the plan of aspect_amd.partition.  Owned entries must equal the serial oracle
on the global mesh; refreshed ghost entries must equal the owner's values."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from aspect_amd import boxmesh as bm  # noqa: E402
from aspect_amd import capi, cases, parallel, partition  # noqa: E402
from oracle import pyoracle  # noqa: E402


class _Standalone:
    """The rank's sub-mesh as a one-rank problem for the serial oracle (all local nodes 'owned')."""

    def __init__(self, loc):
        self.__dict__.update(loc.__dict__)


def compress_and_update(ranks, lev, space, vec, nc):
    """vec: local [n_nodes*nc] partial sums -> owners hold the global sums, ghosts the owner's value."""
    sp = getattr(lev, space)
    so, sn, ro = (getattr(lev, f"send_offsets_{space}"), getattr(lev, f"send_nodes_{space}"),
                  getattr(lev, f"recv_offsets_{space}"))
    v = vec.reshape(-1, nc)
    n_owned = sp.n_owned
    recv = [np.zeros((so[k + 1] - so[k]) * nc) for k in range(len(lev.neighbors))]
    send = [np.ascontiguousarray(v[n_owned + ro[k]:n_owned + ro[k + 1]].reshape(-1)) for k in range(len(lev.neighbors))]
    parallel.exchange_host(ranks.dist, ranks.torch, [(nb, send[k], recv[k]) for k, nb in enumerate(lev.neighbors)])
    for k in range(len(lev.neighbors)):
        np.add.at(v, sn[so[k]:so[k + 1]].astype(np.int64), recv[k].reshape(-1, nc))
    send = [np.ascontiguousarray(v[sn[so[k]:so[k + 1]].astype(np.int64)].reshape(-1)) for k in range(len(lev.neighbors))]
    recv = [np.zeros((ro[k + 1] - ro[k]) * nc) for k in range(len(lev.neighbors))]
    parallel.exchange_host(ranks.dist, ranks.torch, [(nb, send[k], recv[k]) for k, nb in enumerate(lev.neighbors)])
    for k in range(len(lev.neighbors)):
        v[n_owned + ro[k]:n_owned + ro[k + 1]] = recv[k].reshape(-1, nc)
    return v.reshape(-1)


def main():
    ranks = parallel.Ranks(backend="gloo")
    assert ranks.world == 2
    g = cases.nsinker_box(2, n_sinkers=4, ratio=1e4, averaging=bm.AVG_NONE)
    g.reps, g.extents = [2, 1, 1], [2.0, 1.0, 1.0]
    g.build()
    loc = partition.LocalProblem(g, ranks.rank, ranks.world)
    ora_g = pyoracle.solver(g)
    ora_l = pyoracle.solver(loc)      # no transport: the local sub-mesh as a standalone mesh
    rng = np.random.Generator(np.random.PCG64(3))
    worst = 0.0
    for level in range(g.hier.n_levels):
        gl, ll = g.hier.levels[level], loc.hier.levels[level]
        for which, space, nc in ((capi.OP_A, "v", g.dim), (capi.OP_MASS, "p", 1)):
            n = gl.n_dofs_v if space == "v" else gl.n_dofs_p
            x = rng.uniform(-1, 1, n)
            con_g = gl.constrained_dofs_v() if space == "v" else np.zeros(n, dtype=bool)
            x[con_g] = 0.0
            xl = ll.to_local_v(x) if space == "v" else ll.to_local_p(x)
            part = ora_l.vmult(which, xl, level)            # sums over the rank's own cells only
            con_l = ll.constrained_dofs_v() if space == "v" else np.zeros(xl.size, dtype=bool)
            part[con_l] = 0.0                                # identity rows carry no partial sum
            full = compress_and_update(ranks, ll, space, part, nc)
            full[con_l] = xl[con_l]
            ref = ora_g.vmult(which, x, level)
            ref_l = ll.to_local_v(ref) if space == "v" else ll.to_local_p(ref)
            err = np.linalg.norm(full - ref_l) / np.linalg.norm(ref_l)   # owned AND refreshed ghosts
            worst = max(worst, err)
    worst = ranks.max_over_ranks(worst)
    # the sum all-reduce the library's dot products use
    tot = ranks.sum_over_ranks(float(loc.hier.finest.n_owned_nodes_v))
    ok = worst < 1e-12 and int(tot) == g.hier.finest.n_nodes_v
    if ranks.rank == 0:
        print(f"[dist-cpu] {'ok' if ok else 'FAIL'} worst relative error {worst:.2e}, owned nodes {int(tot)}", flush=True)
    ora_g.close(); ora_l.close()
    ranks.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
