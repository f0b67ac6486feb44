#!/usr/bin/env python3
"""Set-up cost per coefficient change (what the reference redoes every nonlinear iteration):
set_coefficients on every level + compute_diagonals + setup_smoothers (eigenvalue estimates),
with the mesh tables (levels, transfer) already in place.   usage: prof_setup.py <refinement> [kind]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspect_amd import capi, cases  # noqa: E402

r = int(sys.argv[1]) if len(sys.argv) > 1 else 5
kind = sys.argv[2] if len(sys.argv) > 2 else "box"
p = (cases.convection_box(3, r) if kind == "box" else cases.nsinker_box(r)).build()
s = capi.Solver(capi.load(), p)
s.solve(allow_failure=True)                      # builds everything once (incl. transfer tables)
for rep in range(2):
    t0 = time.perf_counter()
    for l in range(p.hier.n_levels):
        s._set_coeff(l, p.eta_levels[l])
    s._set_coeff(capi.ACTIVE, p.eta_active)
    t1 = time.perf_counter()
    s._check(s.lib.compute_diagonals(s.ctx))
    t2 = time.perf_counter()
    s.setup_smoothers()
    t3 = time.perf_counter()
    x, repo, rc = s.solve(allow_failure=True)
    t4 = time.perf_counter()
    print(f"r{r}: set_coefficients {1e3 * (t1 - t0):.1f} ms (host tables -> device), compute_diagonals {1e3 * (t2 - t1):.1f} ms, "
          f"setup_smoothers {1e3 * (t3 - t2):.1f} ms, solve {1e3 * (t4 - t3):.1f} ms ({repo.n_cheap_iterations} its)")
s.close()
