#!/usr/bin/env python3
"""Profiling driver: one warm + one timed block-GMG Stokes solve.
    rocprofv3 --kernel-trace --stats ... -- python3 tools/prof_solve.py 5 box"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspect_amd import capi, cases  # noqa: E402

r = int(sys.argv[1]) if len(sys.argv) > 1 else 5
kind = sys.argv[2] if len(sys.argv) > 2 else "box"
p = cases.convection_box(3, r) if kind == "box" else cases.nsinker_box(r)
p.numbering = os.environ.get("AGMG_NUMBERING", p.numbering)
p.build()
opts = {}
if len(sys.argv) > 3:
    opts["coarse_on_host"] = int(sys.argv[3])
s = capi.Solver(capi.load(), p, **opts)
for i in range(2):
    t = time.perf_counter()
    x, rep, rc = s.solve(allow_failure=True)
    print(f"solve {i}: rc {rc}, {rep.n_cheap_iterations}+{rep.n_expensive_iterations} iterations, "
          f"V-cycles A {rep.n_A_preconditioner_cheap} S {rep.n_S_preconditioner_cheap}, setup {rep.setup_seconds * 1e3:.2f} ms, "
          f"solve {rep.solve_seconds * 1e3:.2f} ms, wall {1e3 * (time.perf_counter() - t):.2f} ms")
s.close()
