"""Worker of the distributed parity tests (launched by tests/test_distributed.py through
torch.distributed.run; not collected by pytest).

Every rank owns one subdomain of the same global problem, runs the HIP library on its local tables
with the ghost exchange / all-reduce going through aspect_amd.parallel.Transport, and compares the
OWNED entries of every result with the serial CPU oracle applied to the GLOBAL problem.

usage: dist_worker.py <case> <backend: gloo|nccl|gloo-dev> [what: all|ops|vmult]
All ranks may share one GPU (gloo: the library stages messages through host memory; gloo-dev: the
library hands out device pointers as it does for RCCL and the transport stages them itself)."""
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

from aspect_amd import boxmesh as bm  # noqa: E402
from aspect_amd import capi, cases, parallel, partition  # noqa: E402
from oracle import pyoracle  # noqa: E402

RTOL = 1e-12


def relerr(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def split_replicated(case):
    """'<case>_rep<k>': the levels 0..k-1 are held whole by every rank (agmg_level_desc::replicated)."""
    m = re.match(r"(.*)_rep(\d+)$", case)
    return (m.group(1), int(m.group(2))) if m else (case, 0)


def global_problem(case, world, build=True):
    reps = {1: [1, 1, 1], 2: [2, 1, 1], 4: [2, 2, 1], 8: [2, 2, 2]}[world]
    case = split_replicated(case)[0]
    if case.endswith("_ranklocal"):   # partition.RankLocalProblem: every rank builds its own subdomain only
        case = case[:-len("_ranklocal")]
    if case == "nsinker":
        p = cases.nsinker_box(2, n_sinkers=4, ratio=1e4, averaging=bm.AVG_NONE)
    elif case == "nsinker_idr":   # SolverIDR(2) as the cheap solver (tests/nsinker_gmg_idr.prm)
        p = cases.nsinker_box(2, n_sinkers=4, ratio=1e4, krylov="idr", tol=1e-7, n_cheap=1000, n_expensive=1000)
    elif case == "free_surface":  # free top and x+ boundaries carrying the stabilisation face term
        p = cases.free_surface_blob(3, 2, side_surface=True)
    elif case == "nsinker_harmonic":
        p = cases.nsinker_box(2, n_sinkers=4, ratio=1e4)
    elif case == "box3d":
        p = cases.convection_box(3, 2)
    elif case == "box2d":
        p = cases.convection_box(2, 3)
        reps = {1: [1, 1], 2: [2, 1], 4: [2, 2]}[world]
    elif case == "deformed":
        p = cases.deformed_sinker_box(3, 2)
    elif case == "nsinker_r3":    # several blocks per rank: interior blocks overlap the ghost import
        p = cases.nsinker_box(3, n_sinkers=4, ratio=1e4)
    elif case == "box3d_r3":
        p = cases.convection_box(3, 3)
    elif case == "box3d_r4":      # BASELINE configs[4] in small: one r4 cube per rank
        p = cases.convection_box(3, 4)
    elif case == "shell":         # BASELINE configs[3]: 96 trees of the spherical shell over the ranks
        return cases.nsinker_shell(1, n_cheap=40).build()
    else:
        raise ValueError(case)
    p.reps = reps
    if case != "deformed":   # the deformation map is written for the unit box
        p.extents = [float(r) for r in reps]
    return p.build() if build else p


def main():
    case, backend = sys.argv[1], sys.argv[2]
    what = sys.argv[3] if len(sys.argv) > 3 else "all"
    n_gpu = torch.cuda.device_count()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank % max(n_gpu, 1))
    torch.cuda.set_device(dev)
    ranks = parallel.Ranks(backend=backend.split("-")[0], device=dev)
    tr = parallel.Transport(ranks, device_buffers=1 if backend == "gloo-dev" else None)
    g = global_problem(case, ranks.world)
    case, n_rep = split_replicated(case)
    if case.endswith("_ranklocal"):
        # the rank's tables in closed form from its own subdomain (no global mesh on the product side; the global
        # problem above is the serial oracle's)
        loc = partition.RankLocalProblem(global_problem(case, ranks.world, build=False), ranks.rank, ranks.world,
                                         n_replicated=n_rep, gather=ranks.gather_concat)
    else:
        loc = partition.LocalProblem(g, ranks.rank, ranks.world, n_replicated=n_rep)
    hip = capi.Solver(capi.load(), loc, transport=tr, device=dev.index)
    ora = pyoracle.solver(g)
    fails = []

    def check(name, err, tol):
        if not (err < tol):
            fails.append(f"{name}: {err:.3e} >= {tol:.1e}")

    rng = np.random.Generator(np.random.PCG64(7))
    nl = g.hier.n_levels
    # operators on every level
    for level in range(nl):
        gl, ll = g.hier.levels[level], loc.hier.levels[level]
        xu = rng.uniform(-1, 1, gl.n_dofs_v)
        xu[gl.constrained_dofs_v()] = 0.0
        xp = rng.uniform(-1, 1, gl.n_dofs_p)
        lo, go = ll.owned_dofs_v()
        check(f"A L{level}", relerr(hip.vmult(capi.OP_A, ll.to_local_v(xu), level)[lo], ora.vmult(capi.OP_A, xu, level)[go]), RTOL)
        lo, go = ll.owned_dofs_p()
        check(f"M L{level}", relerr(hip.vmult(capi.OP_MASS, ll.to_local_p(xp), level)[lo], ora.vmult(capi.OP_MASS, xp, level)[go]), RTOL)
        lo, go = ll.owned_dofs_v()
        check(f"dinvA L{level}", relerr(hip.inverse_diagonal(capi.OP_A, level)[lo], ora.inverse_diagonal(capi.OP_A, level)[go]), RTOL)
        lo, go = ll.owned_dofs_p()
        check(f"dinvM L{level}", relerr(hip.inverse_diagonal(capi.OP_MASS, level)[lo], ora.inverse_diagonal(capi.OP_MASS, level)[go]), RTOL)
    # Stokes / B^T on the active mesh
    gf = g.hier.finest
    x = rng.uniform(-1, 1, gf.n_dofs_v + gf.n_dofs_p)
    x[:gf.n_dofs_v][gf.constrained_dofs_v()] = 0.0
    lo, go = loc.owned_block()
    check("Stokes", relerr(hip.vmult(capi.OP_STOKES, loc.to_local(x))[lo], ora.vmult(capi.OP_STOKES, x)[go]), RTOL)
    check("BT", relerr(hip.vmult(capi.OP_BT, loc.to_local(x))[lo], ora.vmult(capi.OP_BT, x)[go]), RTOL)
    if what in ("all", "ops"):
        # transfer
        for level in range(1, nl):
            for which in (capi.OP_A, capi.OP_MASS):
                gF, gC, lF, lC = g.hier.levels[level], g.hier.levels[level - 1], loc.hier.levels[level], loc.hier.levels[level - 1]
                vel = which == capi.OP_A
                nf, nc = (gF.n_dofs_v, gC.n_dofs_v) if vel else (gF.n_dofs_p, gC.n_dofs_p)
                fine, coarse = rng.uniform(-1, 1, nf), rng.uniform(-1, 1, nc)
                tl_f = lF.to_local_v if vel else lF.to_local_p
                tl_c = lC.to_local_v if vel else lC.to_local_p
                of = lF.owned_dofs_v() if vel else lF.owned_dofs_p()
                oc = lC.owned_dofs_v() if vel else lC.owned_dofs_p()
                a = hip.prolongate_add(which, level, tl_f(fine), tl_c(coarse))
                b = ora.prolongate_add(which, level, fine, coarse)
                check(f"prolong {which} L{level}", relerr(a[of[0]], b[of[1]]), RTOL)
                a = hip.restrict_add(which, level, tl_c(coarse), tl_f(fine))
                b = ora.restrict_add(which, level, coarse, fine)
                check(f"restrict {which} L{level}", relerr(a[oc[0]], b[oc[1]]), RTOL)
        # smoothers, V-cycles, preconditioner (the oracle keeps level 0 on the host exactly;
        # the distributed library runs the same Chebyshev "solver" on the device)
        hip.setup_smoothers()
        ora.setup_smoothers()
        for level in range(nl):
            for which in (capi.OP_A, capi.OP_MASS):
                ga, gb = hip.eigenvalue_estimates(which, level)
                oa, ob = ora.eigenvalue_estimates(which, level)
                # level 0 runs 100 CG steps on a tiny system (a converged Lanczos process amplifies the
                # different summation order of the distributed dot products)
                tol = 1e-8 if level > 0 else 1e-6
                check(f"eig max {which} L{level}", abs(gb - ob) / abs(ob), tol)
                check(f"eig min {which} L{level}", abs(ga - oa) / abs(oa), tol)
        fin = loc.hier.finest
        for which in (capi.OP_A, capi.OP_MASS):
            vel = which == capi.OP_A
            src = rng.uniform(-1, 1, gf.n_dofs_v if vel else gf.n_dofs_p)
            if vel:
                src[gf.constrained_dofs_v()] = 0.0
            lo_, go_ = fin.owned_dofs_v() if vel else fin.owned_dofs_p()
            a = hip.vcycle(which, fin.to_local_v(src) if vel else fin.to_local_p(src))
            check(f"vcycle {which}", relerr(a[lo_], ora.vcycle(which, src)[go_]), 1e-9)
        check("precondition", relerr(hip.precondition(loc.to_local(x))[lo], ora.precondition(x)[go]), 1e-9)
    if what == "ops" and case == "shell":
        # configs[3] also solves: 40 cheap iterations (the count to the deck's tolerance is not pinned on this
        # geometry, tests/test_oracle_goldens.py::test_shell_operators_dense_and_solve), residual history and
        # iterate against the serial oracle's
        fin = loc.hier.finest
        xg, rg, _ = hip.solve(allow_failure=True)
        xo, ro, _ = ora.solve(allow_failure=True)
        n = min(rg.n_history, ro.n_history)
        if not (rg.n_cheap_iterations == ro.n_cheap_iterations == 40 and n >= 40):
            fails.append(f"shell bounded solve: iterations {rg.n_cheap_iterations} vs {ro.n_cheap_iterations}, history {n}")
        hg, ho = np.array(rg.history[:n]), np.array(ro.history[:n])
        check("shell residual history", float(np.max(np.abs(hg - ho) / ho)), 1e-5)
        check("shell residual reduction", float(hg[n - 1] / hg[0]), 0.2)
        lo_, go_ = fin.owned_dofs_v()
        check("shell iterate", relerr(xg[:fin.n_owned_nodes_v * fin.dim], xo[go_]), 1e-5)
    if what == "vmult" and ranks.rank == 0:
        print(f"[dist] {case} world={ranks.world} backend={backend}: level operators, Stokes and B^T (exchanges {tr.n_exchange})", flush=True)
    if what == "ops" and ranks.rank == 0:
        print(f"[dist] {case} world={ranks.world} backend={backend}: operators, transfers, V-cycles, preconditioner "
              f"(exchanges {tr.n_exchange}, all-reduces {tr.n_allreduce})", flush=True)
    if what == "all":
        fin = loc.hier.finest
        # the whole solve
        xg, rg, _ = hip.solve()
        xo, ro, _ = ora.solve()
        # (IDR's short recurrences react to the summation order of the distributed dot products)
        if abs(rg.n_cheap_iterations - ro.n_cheap_iterations) > (3 if g.krylov == "idr" else 1):
            fails.append(f"iterations {rg.n_cheap_iterations} vs {ro.n_cheap_iterations}")
        nu_o = fin.n_owned_nodes_v * fin.dim
        lo_, go_ = fin.owned_dofs_v()
        rhs_norm = np.linalg.norm(np.concatenate([g.rhs_u, g.rhs_p]))
        # the tolerance is T * ||f_u - B^T p0||; for the 1e16-gravity box that difference cancels
        # to rounding level, hence the slack at the scale of ||rhs|| (as in test_gpu_parity.py)
        check("solver tolerance", abs(rg.solver_tolerance - ro.solver_tolerance),
              1e-9 * ro.solver_tolerance + g.tol * 1e-12 * rhs_norm)
        check("solution", relerr(xg[:nu_o], xo[go_]), 1e-4 if rhs_norm < 1e6 else 5e-3)
        if ranks.rank == 0:
            print(f"[dist] {case} replicated levels {n_rep} world={ranks.world} backend={backend}: iterations {rg.n_cheap_iterations} "
                  f"(oracle {ro.n_cheap_iterations}), exchanges {tr.n_exchange}, all-reduces {tr.n_allreduce}", flush=True)
    if case in ("nsinker_r3", "box3d_r3"):
        # deal.II's cell_loop overlaps the ghost import with the interior cells; here: interior blocks
        # (a rank that owns every node it touches -- rank 0 -- imports nothing and has no split)
        n_int, n_ifc, n_over = (int(ranks.max_over_ranks(float(v))) for v in hip.overlap_statistics(nl - 1))
        want = os.environ.get("AGMG_NO_OVERLAP", "0") == "0"
        if want and not (n_int > 0 and n_ifc > 0 and n_over > 0):
            fails.append(f"no overlapped operator application: interior {n_int}, interface {n_ifc}, overlapped {n_over}")
        if not want and n_over != 0:
            fails.append("overlap although AGMG_NO_OVERLAP=1")
        if ranks.rank == 0:
            print(f"[dist] finest level: {n_int} interior + {n_ifc} interface blocks, {n_over} overlapped operator applications", flush=True)
    if os.environ.get("AGMG_BRICKS") == "1":
        # the levels that run the brick kernels on this rank (agmg_get_level_kernel_path)
        on = [l for l in range(nl) if hip.kernel_path(l)[0] > 0]
        if ranks.rank == 0:
            print(f"[dist] brick levels {on}", flush=True)
        if not on:
            fails.append("AGMG_BRICKS=1 but no level runs the brick kernels")
    nfail = ranks.sum_over_ranks(float(len(fails)))
    for f in fails:
        print(f"[dist rank {ranks.rank}] FAIL {f}", flush=True)
    hip.close()
    ora.close()
    ranks.barrier()
    ranks.close()
    sys.exit(1 if nfail else 0)


if __name__ == "__main__":
    main()
