"""Kernel trace target: A-block and Schur V-cycles of configs[1] (convection_box_3d r5), 50 preconditioner applications.
rocprofv3 --kernel-trace --stats -d gpurun_out/vc -- python3 tools/vcycle_profile.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspect_amd import capi, cases
lib = capi.load()
p = cases.convection_box(3, int(os.environ.get("REFINEMENT", "5"))).build()
s = capi.Solver(lib, p, coarse_on_host=int(os.environ.get("COARSE_ON_HOST", "1")))
s.solve(allow_failure=True)
if os.environ.get("DUMP_MAPS"):      # address map of the process, to symbolise a stack trace of the runtime afterwards
    with open(os.environ["DUMP_MAPS"], "w") as f:
        f.write(open("/proc/self/maps").read())
t = s.solver_timings(50)
print({k: round(1e6 * v, 1) for k, v in t.items()})
s.close()
