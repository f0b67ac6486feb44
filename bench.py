#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: Stokes / A-block matvec GDoF/s and block-GMG
Stokes-solve time (3-D Q2-Q1), beside the CPU oracle timed on the host cores.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one StokesOperator::vmult (matrix_free_operators.cc:171-278 + deal.II wrapper) on the
workload's active mesh with all inputs resident in HBM.  value = (velocity + pressure DoFs of the
whole job) / time per step.  N > 1 (weak scaling): the domain is N unit cubes (2x1x1, 2x2x1,
2x2x2), one per rank/GPU, each refined like the N = 1 workload; every vmult updates the ghost
values and compresses the interface sums over RCCL (aspect_amd/parallel.py Transport behind
agmg_set_communicator).  Should the distributed set-up fail on any rank, every rank prints the error
and exits non-zero: no line is printed (a number from independent replicas would not be a scaling
number).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6     # vector FP64 (no MFMA: 3x3 contractions)


def algorithmic_bytes_per_cell(nqv, stokes=False):
    """SURVEY.md 8(d), FP64 + 32-bit indices, compulsory traffic with perfect reuse of shared
    DoFs: vectors 16 B/velocity DoF x 24 DoF/cell = 384; node indices 27 x 4 = 108 (one index
    serves the 3 interleaved components: this build's layout; the 81-index figure would be 324);
    viscosity 8 (cell-constant) or 216 (per-q); Cartesian geometry 32.  Stokes adds 16 B x 1
    pressure DoF + 8 x 4 B indices = 48."""
    b = 384 + 108 + (8 if nqv == 1 else 216) + 32
    return b + (48 if stokes else 0)


A_FLOP_PER_CELL = 6800.0     # ~6.8 kflop/cell, Cartesian


def measured_traffic(refinement, n_cells, kernel_label=""):
    """HBM bytes per A-block vmult from the committed PMC passes (profiles/r3_traffic.json, else the earlier
    rounds': separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of tools/pmc_traffic.sh, gfx950
    correction 2 x FETCH_SIZE, calibrated in profiles/r1_fetch_size_calibration.txt).  bench.py cannot
    run rocprofv3 on itself, so the number is the last committed measurement of this same workload and
    these same kernels, or None."""
    for name in ("r3_traffic.json", "r2_traffic.json", "r1_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                e = json.load(f)["convection_box_3d"].get(str(refinement))
            # (a committed measurement only counts for the kernels it was taken on: when the level runs other kernels
            # than the file names, the number is stale and None is reported instead)
            if e and e["n_cells"] == n_cells:
                names = list(e.get("kernels", {}))
                if kernel_label and names and not any(k in kernel_label for k in names):
                    return None
                return float(e["traffic_bytes"])
        except (OSError, KeyError, ValueError):
            pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--refinement", type=int, default=5, help="configs[1]: convection_box_3d, global refinement 5")
    ap.add_argument("--large-refinement", type=int, default=7,
                    help="extra A-block roofline point on a level that fills the chip (0 = skip)")
    ap.add_argument("--scale-refinement", type=int, default=7,
                    help="N > 1: also time the distributed Stokes vmult and solve with this refinement per GPU, next to "
                         "the same cube on one GPU (0 = skip); `value` stays configs[1] per GPU for every N")
    ap.add_argument("--config3-refinement", type=int, default=6,
                    help="also solve configs[2] (nsinker, 8 sinkers, ratio 1e6) at this refinement (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch

    from aspect_amd.parallel import Ranks
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # rehearsal knob for a one-GPU box: AGMG_BENCH_BACKEND=gloo lets N ranks share cuda:0 (RCCL
    # refuses two ranks on one device); the driver's multi-GPU runs use the default, RCCL
    backend = os.environ.get("AGMG_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks = Ranks(backend=backend, device=dev)
    rank, world = ranks.rank, ranks.world

    from aspect_amd import capi, cases
    lib = capi.load()
    barrier, max_over_ranks = ranks.barrier, ranks.max_over_ranks

    # ---------------------------------------------------------------- workload (configs[1])
    def serial_setup():
        prob = cases.convection_box(3, args.refinement).build()
        solver = capi.Solver(lib, prob, device=local_rank)
        fin = prob.hier.finest
        src_h = np.sin(0.37 * np.arange(fin.n_dofs_v + fin.n_dofs_p))
        src_h[:fin.n_dofs_v][fin.constrained_dofs_v()] = 0.0
        return prob, solver, fin, src_h, fin.n_dofs_v, fin.n_dofs_p

    transport_name = ""
    comm_counts, comm_counts_large = {}, {}
    replicated_levels = {}

    def distributed_setup(refinement=None):
        nonlocal transport_name
        from aspect_amd import partition
        from aspect_amd.parallel import RcclTransport, Transport
        reps = {2: [2, 1, 1], 4: [2, 2, 1], 8: [2, 2, 2]}.get(world)
        if reps is None:
            raise ValueError(f"no subdomain layout for {world} ranks")
        # rank-local set-up: every rank generates its own subdomain (one unit cube of the box of `reps` cubes) in
        # closed form, never the global mesh -- host memory and set-up time per rank do not depend on N
        # (tests/test_host_and_abi.py: the tables equal the partition of the global mesh entry for entry)
        g = cases.convection_box(3, refinement or args.refinement)
        g.reps, g.extents = reps, [float(r) for r in reps]
        # the coarsest levels (at most 32768 cells in the whole mesh) are held whole by every rank: no ghost exchange on
        # them, one sum over ranks where the V-cycle steps onto them (AGMG_REPLICATED=<k> overrides, 0 = every level
        # partitioned as in the reference)
        ref_ = refinement or args.refinement
        n_rep = partition.default_replicated_levels([world * 8 ** l for l in range(ref_ + 1)])
        n_rep = int(os.environ.get("AGMG_REPLICATED", n_rep))
        replicated_levels[ref_] = n_rep
        loc = partition.RankLocalProblem(g, rank, world, n_replicated=n_rep, gather=ranks.gather_concat)
        lf = loc.hier.finest
        src_l = np.sin(0.37 * loc.global_index_vector())          # the same global vector as on one rank
        src_l[:lf.n_dofs_v][lf.constrained_dofs_v()] = 0.0
        own = loc.owned_block()[0]

        def one_vmult(solver):
            s_ = torch.from_numpy(src_l).to(dev)
            d_ = torch.zeros_like(s_)
            if lib.vmult_device(solver.ctx, capi.OP_STOKES, capi.ACTIVE, d_.data_ptr(), s_.data_ptr()):
                raise RuntimeError(lib.last_error().decode())
            torch.cuda.synchronize()
            return d_.cpu().numpy()[own]

        # reference transport: torch.distributed callbacks (RCCL point-to-point through PyTorch); it counts the
        # messages: ghost exchanges and all-reduces per Krylov iteration of a solve (configs[1] only: one more solve)
        tr0 = Transport(ranks)
        solver = capi.Solver(lib, loc, transport=tr0, device=local_rank)
        ref = one_vmult(solver)
        if True:   # (both sizes: one more solve on the counting transport)
            _, _, _ = solver.solve(allow_failure=True)
            e0, a0 = tr0.n_exchange, tr0.n_allreduce
            _, rep_c, _ = solver.solve(allow_failure=True)
            its = max(1, rep_c.n_cheap_iterations + rep_c.n_expensive_iterations)
            (comm_counts if refinement is None else comm_counts_large).update(
                {"gmres_iterations": its, "ghost_exchanges_per_iteration": (tr0.n_exchange - e0) / its,
                 "all_reduces_per_iteration": (tr0.n_allreduce - a0) / its,
                 "replicated_coarse_levels": n_rep, "levels": ref_ + 1})
        transport_name = "torch.distributed callbacks"
        want = os.environ.get("AGMG_TRANSPORT", "rccl" if backend == "nccl" else "torch")
        if want == "rccl":
            # the library's own RCCL transport (no host code per exchange); kept only if every rank
            # reproduces the reference transport's result
            bad = 0.0
            try:
                native = capi.Solver(lib, loc, transport=RcclTransport(ranks), device=local_rank)
                got = one_vmult(native)
                if not np.allclose(got, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max()):
                    bad = 1.0
            except Exception as e:  # noqa: BLE001
                bad = 1.0
                print(f"[bench] rank {rank}: in-library RCCL transport unavailable ({e!r})", file=sys.stderr)
            if max_over_ranks(bad) == 0:
                solver.close()
                solver = native
                transport_name = "in-library RCCL (grouped ncclSend/ncclRecv + ncclAllReduce on the library stream)"
        # whole-job DoFs / N = DoFs per GPU (owned)
        return loc, solver, lf, src_l, loc.n_global_dofs_v / world, loc.n_global_dofs_p / world

    parallelism = "1 rank"
    if world > 1:
        failed, err = 0.0, ""
        try:
            prob, solver, fin, src_h, nu, npd = distributed_setup()
            src = torch.from_numpy(src_h).to(dev)
            dst = torch.empty_like(src)
            if lib.vmult_device(solver.ctx, capi.OP_STOKES, capi.ACTIVE, dst.data_ptr(), src.data_ptr()):
                raise RuntimeError(lib.last_error().decode())
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            failed, err = 1.0, repr(e)
        if max_over_ranks(failed) > 0:
            print(f"[bench] rank {rank}: distributed set-up failed ({err or 'on another rank'}); no result", file=sys.stderr)
            ranks.close()
            sys.exit(3)
        else:
            parallelism = (f"{world} subdomains (one unit cube per GPU), ghost update + compress per vmult and "
                           f"all-reduce per dot product; the {replicated_levels.get(args.refinement, 0)} coarsest multigrid "
                           f"levels replicated on every rank; transport: {transport_name}")
    else:
        prob, solver, fin, src_h, nu, npd = serial_setup()
    n = nu + npd
    src = torch.from_numpy(src_h).to(dev)
    dst = torch.empty_like(src)

    def step():
        rc = lib.vmult_device(solver.ctx, capi.OP_STOKES, capi.ACTIVE, dst.data_ptr(), src.data_ptr())
        if rc:
            raise RuntimeError(lib.last_error().decode())

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    ms_per_step = 1e3 * dt / args.steps
    value = world * n / (dt / args.steps) / 1e9

    # ---------------------------------------------------------------- kernel-level numbers (HIP events
    # on the library's stream, inside agmg_bench_vmult)
    nqv = prob.eta_active.shape[1]
    t_A = solver.bench_vmult(capi.OP_A, capi.ACTIVE, 100)
    t_S = solver.bench_vmult(capi.OP_STOKES, capi.ACTIVE, 100)
    bytes_A = fin.n_cells * algorithmic_bytes_per_cell(nqv)
    def a_block_kernels(s_):
        nb, z, nblk = s_.kernel_path(capi.ACTIVE)
        return (f"k_brickq<A> + k_brick_fix ({nb} bricks of 8 x 8 x {z} cells)" if nb else
                f"k_cells_blk<A> + k_fix ({nblk} blocks)" if nblk else "k_cells<A> + k_pull (slot path)")

    roof = {"kernel": "A-block vmult = " + a_block_kernels(solver) + " (ABlockOperator::vmult)", "bound": "hbm",
            "achieved": bytes_A / t_A / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": bytes_A / t_A / 1e9 / HBM_PEAK_GBS,
            "traffic": measured_traffic(args.refinement, fin.n_cells, a_block_kernels(solver)) if world == 1 else None,
            "algorithmic_bytes": bytes_A,
            "us_per_launch": 1e6 * t_A, "gdofs": nu / t_A / 1e9,
            "fp64_frac": fin.n_cells * A_FLOP_PER_CELL / t_A / 1e12 / FP64_PEAK_TFLOPS,
            "bytes_per_cell": algorithmic_bytes_per_cell(nqv)}

    # ---------------------------------------------------------------- full solve
    x, rep, rc = solver.solve(allow_failure=True)      # first call: tables of the transfers, smoothers
    x, rep2, rc = solver.solve(allow_failure=True)     # smoothers cached
    solve_s = max_over_ranks(rep2.solve_seconds)
    # the reference's own timer sections (local_smoothing.cc:1234-1375) and the smoother set-up that its
    # solve() redoes every time (:918-1058): "Solve Stokes system" of the reference = setup + solve
    sections = solver.solver_timings(5)
    setup_s = max_over_ranks(sections["smoother_setup"])
    extra = {"stokes_vmult_us": 1e6 * t_S, "a_block_vmult_us": 1e6 * t_A,
             "a_block_gdofs": nu / t_A / 1e9,
             "solve_seconds": solve_s, "solve_setup_seconds": setup_s,
             "solve_seconds_including_smoother_setup": solve_s + setup_s,
             "solver_timings_us": {k: 1e6 * max_over_ranks(v) for k, v in sections.items()},
             "gmres_iterations": f"{rep2.n_cheap_iterations}+{rep2.n_expensive_iterations}",
             "vcycles_A": rep2.n_A_preconditioner_cheap, "vcycles_S": rep2.n_S_preconditioner_cheap,
             "solve_status": rc, "velocity_dofs": int(world * nu), "pressure_dofs": int(world * npd)}
    solver.close()

    # ---------------------------------------------------------------- chip-filling level
    roof_large = None
    if args.large_refinement and rank == 0:
        pl = cases.convection_box(3, args.large_refinement)
        pl.build_single_level()
        sl = capi.Solver(lib, pl, device=local_rank, single_level=True)
        fl = pl.hier.finest
        tl = sl.bench_vmult(capi.OP_A, capi.ACTIVE, 20)
        bl = fl.n_cells * algorithmic_bytes_per_cell(pl.eta_active.shape[1])
        roof_large = {"workload": f"convection_box_3d r{args.large_refinement} ({fl.n_cells} cells, {fl.n_dofs_v} velocity DoFs)",
                      "kernel": "A-block vmult = " + a_block_kernels(sl),
                      "achieved": bl / tl / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": bl / tl / 1e9 / HBM_PEAK_GBS, "us_per_launch": 1e6 * tl,
                      "traffic": measured_traffic(args.large_refinement, fl.n_cells, a_block_kernels(sl)), "algorithmic_bytes": bl,
                      "gdofs": fl.n_dofs_v / tl / 1e9,
                      "fp64_frac": fl.n_cells * A_FLOP_PER_CELL / tl / 1e12 / FP64_PEAK_TFLOPS}
        sl.close()

    # ---------------------------------------------------------------- N > 1: a larger cube per GPU (less latency-bound
    # than configs[1]) through the same distributed path, and the same cube on one GPU beside it
    scale_large = None
    if world > 1 and args.scale_refinement:
        failed, err = 0.0, ""
        try:
            loc2, solver2, fin2, src2_h, nu2, np2 = distributed_setup(args.scale_refinement)
            src2 = torch.from_numpy(src2_h).to(dev)
            dst2 = torch.empty_like(src2)

            def step2():
                if lib.vmult_device(solver2.ctx, capi.OP_STOKES, capi.ACTIVE, dst2.data_ptr(), src2.data_ptr()):
                    raise RuntimeError(lib.last_error().decode())
            for _ in range(3):
                step2()
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            for _ in range(20):
                step2()
            torch.cuda.synchronize()
            barrier()
            dt2 = max_over_ranks(time.perf_counter() - t0) / 20
            _, _, _ = solver2.solve(allow_failure=True)
            _, rep_l, rc_l = solver2.solve(allow_failure=True)
            solve2 = max_over_ranks(rep_l.solve_seconds)
            solver2.close()
            # the same cube alone on this rank's GPU
            p1 = cases.convection_box(3, args.scale_refinement).build()
            s1 = capi.Solver(lib, p1, device=local_rank)
            t1 = max_over_ranks(s1.bench_vmult(capi.OP_STOKES, capi.ACTIVE, 20))
            _, _, _ = s1.solve(allow_failure=True)
            _, rep_1, _ = s1.solve(allow_failure=True)
            solve1 = max_over_ranks(rep_1.solve_seconds)
            f1 = p1.hier.finest
            s1.close()
            scale_large = {"workload": f"convection_box_3d r{args.scale_refinement} per GPU ({fin2.n_cells} cells, "
                                       f"{int(nu2)}+{int(np2)} DoFs per GPU)",
                           "stokes_vmult_us": 1e6 * dt2, "gdofs": world * (nu2 + np2) / dt2 / 1e9,
                           "solve_seconds": solve2, "gmres_iterations": f"{rep_l.n_cheap_iterations}+{rep_l.n_expensive_iterations}",
                           "solve_status": rc_l, "messages_per_krylov_iteration": dict(comm_counts_large) or None,
                           "one_gpu_same_cube": {"stokes_vmult_us": 1e6 * t1, "gdofs": (f1.n_dofs_v + f1.n_dofs_p) / t1 / 1e9,
                                                 "solve_seconds": solve1,
                                                 "gmres_iterations": f"{rep_1.n_cheap_iterations}+{rep_1.n_expensive_iterations}"}}
        except Exception as e:  # noqa: BLE001
            failed, err = 1.0, repr(e)
        if max_over_ranks(failed) > 0:
            scale_large = {"error": f"rank {rank}: {err or 'failed on another rank'}"}

    # ---------------------------------------------------------------- configs[2]: nsinker, 8 sinkers, ratio 1e6, r6
    config3 = None
    if args.config3_refinement and rank == 0 and world == 1:
        p3 = cases.nsinker_box(args.config3_refinement, average_density=True).build()
        s3 = capi.Solver(lib, p3, device=local_rank)
        x3, r3a, rc3 = s3.solve(allow_failure=True)
        x3, r3b, rc3 = s3.solve(allow_failure=True)
        t3 = s3.solver_timings(3)
        config3 = {"workload": f"benchmarks/nsinker r{args.config3_refinement}: 8 sinkers, viscosity ratio 1e6, "
                               f"{p3.hier.finest.n_cells} cells, {p3.hier.finest.n_dofs_v}+{p3.hier.finest.n_dofs_p} DoFs",
                   "gmres_iterations": f"{r3b.n_cheap_iterations}+{r3b.n_expensive_iterations}", "solve_status": rc3,
                   "solve_seconds": r3b.solve_seconds, "solve_seconds_including_smoother_setup": r3b.solve_seconds + t3["smoother_setup"],
                   "oracle_iterations_r2_r3_r4": "35, 46, 72 (identical on the GPU; profiles/r2_config3_nsinker_iterations.jsonl)"}
        s3.close()

    # ---------------------------------------------------------------- CPU baseline (rank 0, N = 1)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        # the CPU restatement on the host cores of this box: cell loop threaded colour by colour
        # (OpenMP); the box gives one GPU's share of the host, 16 cores
        cores = max(1, min(16, os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
        os.environ["ORACLE_THREADS"] = str(cores)
        from oracle import pyoracle
        o = pyoracle.solver(prob)
        # the restatement with SIMD across cells (oracle/cpu_simd.cpp: one cell per vector lane, as deal.II's
        # VectorizedArray batches them; AVX-512 = 8 lanes, AVX2 = 4, picked from the host's CPUID), OpenMP
        # threads over the cell batches of a colour; checked here against the scalar oracle first
        ref_cpu = o.vmult(capi.OP_STOKES, src_h)
        got_cpu, lanes = o.vmult_simd(capi.OP_STOKES, src_h)
        cpu_err = float(np.linalg.norm(got_cpu - ref_cpu) / np.linalg.norm(ref_cpu))
        if cpu_err > 1e-12:
            raise RuntimeError(f"SIMD CPU baseline differs from the oracle: {cpu_err:.3e}")
        # the reference's harness protocol (local_smoothing.cc:1236-1264): 1 warm-up, 10 timings x 10 repeats, mean
        times = []
        t_begin = time.perf_counter()
        for _ in range(10):
            times.append(o.bench_vmult_simd(capi.OP_STOKES, src_h, 10)[0])   # (timed inside the library: the vmult alone)
            if time.perf_counter() - t_begin > 25.0:
                break
        tc = float(np.mean(times))
        # the scalar test oracle beside it (3 repeats)
        t0 = time.perf_counter()
        for _ in range(3):
            o.vmult(capi.OP_STOKES, src_h)
        tc_scalar = (time.perf_counter() - t0) / 3
        cpu = {"value": n / tc / 1e9, "unit": "GDoF/s", "cores": cores, "kind": "port", "restatement": "SIMD across cells (one cell per vector lane, as deal.II's VectorizedArray)", "vectorised": True,
               "simd_lanes": lanes, "rel_err_vs_oracle": cpu_err, "scalar_oracle_gdofs": n / tc_scalar / 1e9,
               "sample": f"{len(times)} x 10 Stokes vmults of the same r{args.refinement} workload by oracle/cpu_simd.cpp: a "
                         f"restatement with SIMD across cells ({lanes} cells per vector, sum-factorised, Cartesian), "
                         f"OpenMP over the cell batches of a colour, {cores} threads; the mean of the timings as in "
                         f"the reference's harness. A reported baseline: the ratio to it says nothing about kernel "
                         f"quality"}
        o.close()

    if rank == 0:
        line = {"metric": "stokes_matvec_gdofs", "value": value, "unit": "GDoF/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic",
                "config": {"workload": f"cookbooks/convection_box_3d, global refinement {args.refinement}, "
                                       f"{fin.n_cells} cells, {int(nu)}+{int(npd)} DoFs per GPU, per-q viscosity tables",
                           "parallelism": parallelism},
                "roofline": roof, "cpu_baseline": cpu, "roofline_large": roof_large, "config3_nsinker": config3, "weak_scaling_larger_cube": scale_large, "messages_per_krylov_iteration": comm_counts or None,
                **extra}
        print(json.dumps(line))
    ranks.close()


if __name__ == "__main__":
    main()
