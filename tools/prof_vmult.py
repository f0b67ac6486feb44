#!/usr/bin/env python3
"""Profiling driver: N A-block (and Stokes / mass) vmults on one big level, for rocprofv3.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o vm -- python3 tools/prof_vmult.py 7 20"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspect_amd import capi, cases  # noqa: E402

r = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
kind = sys.argv[3] if len(sys.argv) > 3 else "box"
if kind == "box":
    p = cases.convection_box(3, r)       # per-q viscosity tables
else:
    p = cases.nsinker_box(r)             # cell-constant (harmonic) viscosity
p.numbering = os.environ.get("AGMG_NUMBERING", p.numbering)
p.build_single_level()
s = capi.Solver(capi.load(), p, single_level=True)
ops = ((capi.OP_A, "A"), (capi.OP_STOKES, "Stokes"), (capi.OP_MASS, "mass"))
if len(sys.argv) > 4:
    ops = [o for o in ops if o[1] in sys.argv[4].split(",")]
for which, name in ops:
    # the Schur smoother applies the LEVEL operators' mass matrices (assembled rows); level 0 = the only level here
    t = s.bench_vmult(which, 0 if which == capi.OP_MASS else capi.ACTIVE, n)
    nd = s.size(which)
    print(f"{name}: {1e6 * t:.1f} us per vmult, {nd / t / 1e9:.2f} GDoF/s")
s.close()
