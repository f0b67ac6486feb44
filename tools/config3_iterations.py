#!/usr/bin/env python3
"""BASELINE configs[2] (benchmarks/nsinker: 8 sinkers, viscosity ratio 1e6, harmonic average of
viscosity and density as its deck asks, tolerance 1e-6, GMRES(100), 200 cheap steps): GMRES iteration counts and solve times
of the HIP library up to r = 6 beside the CPU oracle's up to the refinement it solves in about a minute.
    python tools/config3_iterations.py [max_gpu_r] [max_oracle_r]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspect_amd import capi, cases  # noqa: E402
from oracle import pyoracle  # noqa: E402

max_gpu = int(sys.argv[1]) if len(sys.argv) > 1 else 6
max_ora = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = []
for r in range(2, max_gpu + 1):
    p = cases.nsinker_box(r, average_density=True).build()
    g = capi.Solver(capi.load(), p)
    x, rep, rc = g.solve(allow_failure=True)
    x, rep2, rc2 = g.solve(allow_failure=True)       # second solve: smoothers cached
    row = {"refinement": r, "cells": p.hier.finest.n_cells, "velocity_dofs": p.hier.finest.n_dofs_v,
           "eta_min": p.eta_min, "eta_max": p.eta_max, "gpu_status": rc, "gpu_iterations": f"{rep.n_cheap_iterations}+{rep.n_expensive_iterations}",
           "gpu_vcycles_A": rep.n_A_preconditioner_cheap, "gpu_final_over_tol": rep.final_linear_residual / rep.solver_tolerance,
           "gpu_solve_seconds_with_setup": rep.solve_seconds + rep.setup_seconds, "gpu_solve_seconds_cached": rep2.solve_seconds}
    g.close()
    if r <= max_ora:
        t = time.time()
        o = pyoracle.solver(p)
        xo, ro, rco = o.solve(allow_failure=True)
        o.close()
        row.update(oracle_status=rco, oracle_iterations=f"{ro.n_cheap_iterations}+{ro.n_expensive_iterations}", oracle_seconds=time.time() - t)
    rows.append(row)
    print(json.dumps(row), flush=True)
