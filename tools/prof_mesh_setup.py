import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aspect_amd import capi, cases
import ctypes as C
r = int(sys.argv[1])
t0 = time.perf_counter()
p = cases.convection_box(3, r).build()
t1 = time.perf_counter()
print(f"python host mesh+tables r{r}: {t1 - t0:.2f} s", flush=True)
lib = capi.load()
orig_set_level = capi.Solver._set_level
tl = [0.0]
def timed(self, l, lev):
    a = time.perf_counter(); orig_set_level(self, l, lev); tl[0] += time.perf_counter() - a
capi.Solver._set_level = timed
a = time.perf_counter()
s = capi.Solver(lib, p)
b = time.perf_counter()
print(f"Solver() total {b - a:.2f} s of which set_level (all levels) {tl[0]:.2f} s", flush=True)
a = time.perf_counter(); s.setup_smoothers(); b = time.perf_counter()
print(f"first setup_smoothers (transfer tables + estimates) {b - a:.2f} s", flush=True)
