#!/usr/bin/env python3
"""BASELINE configs[3] on one GPU: A-block / Stokes vmult (general Jacobians) and the solve of
benchmarks/nsinker_spherical_shell at a uniform refinement r (96 * 8^r cells).
    python tools/prof_shell.py [r] [n_repeat]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspect_amd import capi, cases  # noqa: E402

r = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
t0 = time.time()
p = cases.nsinker_shell(r).build()
fin = p.hier.finest
s = capi.Solver(capi.load(), p)
tA = s.bench_vmult(capi.OP_A, capi.ACTIVE, n)
tS = s.bench_vmult(capi.OP_STOKES, capi.ACTIVE, n)
x, rep, rc = s.solve(allow_failure=True)
# general Jacobian, cell-constant viscosity: 384 (vectors) + 108 (node indices, one per node) + 8 + 2160 B/cell
bytes_cell = 384 + 108 + 8 + 2160
print(json.dumps({"workload": f"nsinker_spherical_shell r{r}: {fin.n_cells} cells, {fin.n_dofs_v}+{fin.n_dofs_p} DoFs",
                  "host_mesh_seconds": time.time() - t0, "eta_range": [p.eta_min, p.eta_max], "pressure_scaling": p.pressure_scaling,
                  "a_block_vmult_us": 1e6 * tA, "a_block_gdofs": fin.n_dofs_v / tA / 1e9,
                  "stokes_vmult_us": 1e6 * tS, "bytes_per_cell": bytes_cell,
                  "a_block_hbm_frac": fin.n_cells * bytes_cell / tA / 8e12,
                  "solve_status": rc, "gmres_iterations": f"{rep.n_cheap_iterations}+{rep.n_expensive_iterations}",
                  "final_residual_over_tolerance": rep.final_linear_residual / rep.solver_tolerance,
                  "solve_seconds": rep.solve_seconds, "setup_seconds": rep.setup_seconds}))
s.close()
