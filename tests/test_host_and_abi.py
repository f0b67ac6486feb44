"""CPU suite, part 2: the C-ABI library loads and exports every symbol include/agmg.h declares
(no compute without a GPU), refuses to run without a device instead of falling back, the host
table generator is consistent, and the rank plumbing works at world_size 2 over gloo."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from aspect_amd import boxmesh as bm
from aspect_amd import capi, cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "agmg.h")).read()
    return sorted(set(re.findall(r"\b(agmg_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(capi.product_library_path())
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libaspect_amd.so lacks {n}"
    # and the ctypes layer binds the same set
    assert sorted("agmg_" + s for s in capi.SYMBOLS) == names


def test_struct_layouts_match_header():
    lib = capi.load()
    opt = capi.Options()
    lib.default_options(C.byref(opt))
    assert (opt.dim, opt.degree_v, opt.n_levels) == (3, 2, 1)
    assert opt.linear_tolerance == 1e-7 and opt.gmres_restart == 100
    assert opt.n_cheap_steps == 1000 and opt.n_expensive_steps == 1000
    assert (opt.smoother_degree, opt.smoothing_range, opt.eig_cg_n_iterations) == (4, 15.0, 10)
    assert (opt.coarse_degree, opt.coarse_range, opt.coarse_eig_cg_n_iterations) == (8, 1e-3, 100)
    assert opt.schur_smoother_steps == 4 and opt.coarse_on_host == 1 and opt.world_size == 1


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback_without_gpu():
    lib = capi.load()
    opt = capi.Options()
    lib.default_options(C.byref(opt))
    assert not lib.create(C.byref(opt))
    assert b"no HIP device" in lib.last_error()
    with pytest.raises(capi.AgmgError):
        capi.Solver(lib, cases.nsinker_box(1, n_sinkers=4, ratio=1e4).build())


def test_box_hierarchy_tables():
    h = bm.BoxHierarchy(3, 2, bc=[bm.BC_ZERO] * 6)
    fin = h.finest
    assert (fin.n_cells, fin.n_dofs_v, fin.n_dofs_p) == (64, 2187, 125)
    # children of cell P on level l-1 are cells 8P..8P+7 on level l (Morton order)
    for l in range(1, 3):
        c, f = h.levels[l - 1], h.levels[l]
        cc = c.cell_centers()
        fc = f.cell_centers().reshape(c.n_cells, 8, 3)
        assert np.all(np.abs(fc.mean(axis=1) - cc) < 1e-14)
    # every boundary node of a zero-velocity box carries all three constraint bits
    x = fin.node_coords_v()
    on_bdry = np.any((x == 0) | (x == 1), axis=1)
    assert np.all((fin.cmask_v == 7) == on_bdry)
    # free slip: only the normal component
    fs = bm.BoxHierarchy(3, 1).finest
    x = fs.node_coords_v()
    assert np.all(((fs.cmask_v >> 2) & 1).astype(bool) == ((x[:, 2] == 0) | (x[:, 2] == 1)))


def test_viscosity_tables_match_reference_rules():
    p = cases.nsinker_box(2, n_sinkers=4, ratio=1e4).build()
    assert p.eta_active.shape == (64, 1)                       # DGQ0: second extent 1
    # level tables: arithmetic mean of the 2^dim children (FE_DGQ(0) restriction)
    assert np.allclose(p.eta_levels[1][:, 0], p.eta_levels[2][:, 0].reshape(8, 8).mean(axis=1))
    q = cases.convection_box(3, 2).build()
    assert q.eta_active.shape == (64, 27)                      # project to Q1: per-q tables
    assert all(t.shape[1] == 27 for t in q.eta_levels)
    assert q.pressure_scaling == 100.0


WORKER = r"""
import os, sys
sys.path.insert(0, %r)
from aspect_amd.parallel import Ranks, aggregate_throughput
r = Ranks(backend="gloo")
assert r.world == 2
t = r.max_over_ranks(1.0 + r.rank)
s = r.sum_over_ranks(10.0 * (r.rank + 1))
r.barrier()
assert t == 2.0 and s == 30.0, (t, s)
v = aggregate_throughput(100.0, 5, t, r.world)
assert v == 2 * 100.0 * 5 / 2.0
r.close()
print("rank", r.rank, "ok")
"""


def test_rank_plumbing_world_size_2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()


def test_simd_cpu_baseline_matches_the_oracle(oracle_lib):
    """bench.py's cpu_baseline leg (oracle/cpu_simd.cpp, SIMD across cells) computes the oracle's operators:
    A, Stokes and B^T on Cartesian 3-D levels with per-q and cell-constant viscosity, compressible too."""
    import numpy as np
    from aspect_amd import capi, cases
    for make, comp in ((lambda: cases.convection_box(3, 2), 0), (lambda: cases.nsinker_box(2, n_sinkers=4, ratio=1e4), 1)):
        p = make().build()
        o = capi.Solver(oracle_lib, p)
        try:
            if comp:
                for l in range(p.hier.n_levels):
                    o._set_coeff(l, p.eta_levels[l], is_compressible=1)
                o._set_coeff(capi.ACTIVE, p.eta_active, is_compressible=1)
            rng = np.random.Generator(np.random.PCG64(3))
            for which in (capi.OP_A, capi.OP_STOKES, capi.OP_BT):
                src = rng.uniform(-1, 1, o.size(which))
                a = o.vmult(which, src)
                b, lanes = o.vmult_simd(which, src)
                assert lanes in (4, 8)
                assert np.linalg.norm(a - b) <= 1e-13 * np.linalg.norm(a), which
        finally:
            o.close()


@pytest.mark.parametrize("refinement,reps,size", [(2, [2, 1, 1], 2), (1, [2, 2, 2], 8), (2, [2, 2, 2], 2), (2, [4, 1, 1], 2)])
def test_rank_local_setup_equals_partition_of_the_global_mesh(refinement, reps, size):
    """partition.RankLocalProblem (each rank generates its own subdomain in closed form, BASELINE configs[4]'s
    set-up) against partition.LocalProblem cut out of the global mesh: every table of every level entry for entry
    (numbering, ownership, exchange plans, constraint masks), the coefficient tables bit for bit, the right-hand
    side to rounding (own cells plus a halo layer instead of the global assembly: another order of the sums)."""
    from aspect_amd import partition

    def case():
        c = cases.nsinker_box(refinement, n_sinkers=4, ratio=1e4)
        c.reps, c.extents = reps, [float(x) for x in reps]
        return c

    g = case().build()
    keys = ("cells", "n_cells", "n_nodes_v", "n_owned_nodes_v", "n_nodes_p", "n_owned_nodes_p", "cell_nodes_v", "cell_nodes_p",
            "cmask_v", "cmask_p", "cell_h", "neighbors", "send_offsets_v", "send_nodes_v", "recv_offsets_v", "send_offsets_p",
            "send_nodes_p", "recv_offsets_p", "global_nodes_v", "global_nodes_p", "n_global_nodes_v", "n_global_nodes_p")
    for rank in range(size):
        a, b = partition.LocalProblem(g, rank, size), partition.RankLocalProblem(case(), rank, size)
        assert a.hier.n_levels == b.hier.n_levels
        for la, lb in zip(a.hier.levels, b.hier.levels):
            for k in keys:
                assert np.array_equal(np.asarray(getattr(la, k)), np.asarray(getattr(lb, k))), (rank, la.level, k)
        for ta, tb in zip(a.eta_levels + [a.eta_active], b.eta_levels + [b.eta_active]):
            assert np.array_equal(ta, tb)
        assert np.allclose(a.rhs_u, b.rhs_u, rtol=1e-13, atol=1e-13 * np.abs(a.rhs_u).max())
        assert np.array_equal(a.p0, b.p0) and np.array_equal(a.rhs_p, b.rhs_p)
        x = np.arange(g.hier.finest.n_dofs_v + g.hier.finest.n_dofs_p, dtype=np.float64)
        assert np.array_equal(a.to_local(x), b.to_local(x)) and np.array_equal(b.to_local(x), b.global_index_vector())
        assert all(np.array_equal(p, q) for p, q in zip(a.owned_block(), b.owned_block()))


@pytest.mark.parametrize("averaging", ["harmonic", "project_q1"])
def test_replicated_coarse_levels_of_the_host_partitions(averaging):
    """agmg_level_desc::replicated on the host side: the levels 0..k-1 of both partitions (cut out of the global mesh /
    generated per rank) are the global levels themselves, without an exchange plan; the first distributed level names
    its parents in the whole coarser level; the coefficient tables of the replicated levels, which the rank-local
    set-up forms from a gather of the coarsest distributed level, equal the global ones bit for bit."""
    from aspect_amd import boxmesh as bm
    from aspect_amd import partition
    size, reps, k = 4, [2, 2, 1], 2

    def case():
        c = cases.nsinker_box(3, n_sinkers=4, ratio=1e4, averaging=bm.AVG_HARMONIC if averaging == "harmonic" else bm.AVG_PROJECT_Q1)
        c.reps, c.extents = reps, [float(x) for x in reps]
        return c

    g = case().build()
    assert partition.default_replicated_levels([lv.n_cells for lv in g.hier.levels], max_cells=300) == 3
    assert partition.default_replicated_levels([lv.n_cells for lv in g.hier.levels], max_cells=10 ** 9) == g.hier.n_levels - 1
    for rank in range(size):
        a = partition.LocalProblem(g, rank, size, n_replicated=k)
        for l in range(g.hier.n_levels):
            la, gl = a.hier.levels[l], g.hier.levels[l]
            assert la.replicated == (l < k)
            if l < k:
                assert la.n_cells == gl.n_cells and la.n_owned_nodes_v == la.n_nodes_v == gl.n_nodes_v and not la.neighbors
                assert np.array_equal(la.cell_nodes_v, gl.cell_nodes_v) and np.array_equal(la.cmask_v, gl.cmask_v)
                assert np.array_equal(a.eta_levels[l], g.eta_levels[l])
            if l == k:
                assert np.array_equal(la.parent_cell, la.cells // 8) and np.array_equal(la.child_index, la.cells % 8)
                assert la.parent_cell.max() < g.hier.levels[l - 1].n_cells
            else:
                assert not hasattr(la, "parent_cell")
    # rank-local set-up with a real gather: all ranks in lock step (threads stand in for the processes)
    import threading
    results, slots, barrier = [None] * size, {}, threading.Barrier(size)

    def worker(rank):
        calls = [0]

        def gather(arr):
            key = calls[0]
            calls[0] += 1
            slots[(key, rank)] = np.array(arr, dtype=np.float64)
            barrier.wait()
            out = np.concatenate([slots[(key, r)] for r in range(size)], axis=0)
            barrier.wait()
            return out
        results[rank] = partition.RankLocalProblem(case(), rank, size, n_replicated=k, gather=gather)

    th = [threading.Thread(target=worker, args=(r,)) for r in range(size)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for rank in range(size):
        b, a = results[rank], partition.LocalProblem(g, rank, size, n_replicated=k)
        assert b is not None
        for l in range(g.hier.n_levels):
            la, lb = a.hier.levels[l], b.hier.levels[l]
            for key in ("n_cells", "n_nodes_v", "n_owned_nodes_v", "n_nodes_p", "cell_nodes_v", "cell_nodes_p", "cmask_v", "neighbors"):
                assert np.array_equal(np.asarray(getattr(la, key)), np.asarray(getattr(lb, key))), (rank, l, key)
            assert lb.replicated == (l < k)
            if l == k:
                assert np.array_equal(la.parent_cell, lb.parent_cell) and np.array_equal(la.child_index, lb.child_index)
            assert np.array_equal(a.eta_levels[l], b.eta_levels[l]), (rank, l)


def test_symmetric_exchange_plans_agree_between_ranks():
    """agmg_level_desc::sym_*: every pair of ranks that touch a common pressure node list their shared nodes in the same
    order (ascending global index), neighbours ascending by rank, on every level; the rank-local set-up gives the same
    plan as the partition of the global mesh; a rank's plan covers exactly its ghost nodes and its exported nodes."""
    from aspect_amd import partition
    size, reps = 8, [2, 2, 2]

    def case():
        c = cases.nsinker_box(2, n_sinkers=4, ratio=1e4)
        c.reps, c.extents = reps, [float(x) for x in reps]
        return c

    g = case().build()
    loc = [partition.LocalProblem(g, r, size) for r in range(size)]
    for l in range(g.hier.n_levels):
        L = [p.hier.levels[l] for p in loc]
        for a in range(size):
            nb = list(L[a].sym_neighbors_p)
            assert nb == sorted(nb) and a not in nb
            shared = set()
            for i, k in enumerate(nb):
                mine = L[a].sym_nodes_p[L[a].sym_offsets_p[i]:L[a].sym_offsets_p[i + 1]]
                j = list(L[k].sym_neighbors_p).index(a)
                theirs = L[k].sym_nodes_p[L[k].sym_offsets_p[j]:L[k].sym_offsets_p[j + 1]]
                ga, gb = L[a].p.l2g[mine], L[k].p.l2g[theirs]
                assert np.array_equal(ga, gb) and np.all(np.diff(ga) > 0)
                shared |= set(int(x) for x in mine)
            exported = set(int(x) for x in L[a].send_nodes_p)
            ghosts = set(range(L[a].n_owned_nodes_p, L[a].n_nodes_p))
            assert shared == exported | ghosts
            b = partition.RankLocalProblem(case(), a, size).hier.levels[l]
            for key in ("sym_neighbors_p", "sym_offsets_p", "sym_nodes_p", "sym_neighbors_v", "sym_offsets_v", "sym_nodes_v"):
                assert np.array_equal(getattr(L[a], key), getattr(b, key)), (l, a, key)


def test_ctypes_mirrors_have_the_layout_of_the_header(tmp_path):
    """include/agmg.h against the ctypes mirrors of aspect_amd/capi.py: size of every struct that crosses the ABI and the
    offsets of the fields added last (a drift between the two would pass garbage pointers to the library)."""
    import ctypes as C
    import subprocess
    src = tmp_path / "layout.c"
    src.write_text(r'''
#include <stddef.h>
#include <stdio.h>
#include "agmg.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu ", sizeof(agmg_options), sizeof(agmg_level_desc), sizeof(agmg_cell_data), sizeof(agmg_face_data),
         sizeof(agmg_solve_report), sizeof(agmg_comm));
  printf("%zu %zu %zu %zu %zu\n", offsetof(agmg_level_desc, n_hanging_v), offsetof(agmg_level_desc, n_copy_v),
         offsetof(agmg_level_desc, replicated), offsetof(agmg_level_desc, sym_nodes_p), offsetof(agmg_level_desc, sym_nodes_v));
  return 0;
}
''')
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-I", inc, str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    from aspect_amd.parallel import CommStruct
    want = [C.sizeof(capi.Options), C.sizeof(capi.LevelDesc), C.sizeof(capi.CellData), C.sizeof(capi.FaceData),
            C.sizeof(capi.SolveReport), C.sizeof(CommStruct),
            capi.LevelDesc.n_hanging_v.offset, capi.LevelDesc.n_copy_v.offset, capi.LevelDesc.replicated.offset,
            capi.LevelDesc.sym_nodes_p.offset, capi.LevelDesc.sym_nodes_v.offset]
    assert got == want, (got, want)
