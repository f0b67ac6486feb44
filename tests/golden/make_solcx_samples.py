#!/usr/bin/env python3
"""Tabulates the analytic SolCx solution (velocity, pressure) at the quadrature points the
reference's error postprocessor uses (benchmarks/solcx/solcx.h:3129-3134: QGauss(degree_v + 2) = 4
points per direction on every cell) for the 4 x 4 and 16 x 16 meshes of tests/sol_cx_2_gmg and
tests/sol_cx_4_gmg, and writes the NUMBERS to tests/golden/solcx_analytic_r{2,4}.npy
(columns x, z, u_x, u_z, p).  Run in the build container, where /root/reference exists: the analytic
function `_Velic_solCx` (solcx.h:46-2890, plain C apart from numbers::PI) is compiled from the
reference's own file where it lies, into a temporary directory; no source is copied into this repo.
    python tests/golden/make_solcx_samples.py"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def build(tmp):
    lines = open(os.path.join(REF, "benchmarks", "solcx", "solcx.h")).read().split("\n")
    start = next(i for i, l in enumerate(lines) if "void _Velic_solCx(" in l)
    end = next(i for i in range(start, len(lines)) if "The exact solution for the SolCx benchmark" in lines[i])
    # the function's closing brace is the last "}" before the doc comment that follows it
    while "}" not in lines[end]:
        end -= 1
    body = "\n".join(lines[start:end + 1])
    src = os.path.join(tmp, "velic.cpp")
    with open(src, "w") as f:
        f.write("#include <cmath>\n#include <cassert>\nnamespace numbers { const double PI = 3.141592653589793238462643383279502884; }\n")
        f.write(body)
        f.write("""
extern "C" void solcx_eval(int n, const double *xz, double eta_B, double *out) {
  for (int i = 0; i < n; ++i) {
    double pos[2] = {xz[2 * i], xz[2 * i + 1]}, vel[2], p, ts[3], sr[3];
    _Velic_solCx(pos, 1.0, eta_B, 0.5, 1, vel, &p, ts, sr);
    out[3 * i] = vel[0]; out[3 * i + 1] = vel[1]; out[3 * i + 2] = p;
  }
}
""")
    so = os.path.join(tmp, "velic.so")
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-o", so, src])
    return C.CDLL(so)


def main():
    gx, _ = np.polynomial.legendre.leggauss(4)
    gx = 0.5 * (gx + 1.0)
    with tempfile.TemporaryDirectory() as tmp:
        lib = build(tmp)
        lib.solcx_eval.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]
        for r in (2, 4):
            n = 1 << r
            h = 1.0 / n
            pts = []
            for cz in range(n):            # cells lexicographic (x fastest), q-points lexicographic (x fastest)
                for cx in range(n):
                    for qz in range(4):
                        for qx in range(4):
                            pts.append(((cx + gx[qx]) * h, (cz + gx[qz]) * h))
            xz = np.ascontiguousarray(pts, dtype=np.float64)
            out = np.zeros((xz.shape[0], 3))
            lib.solcx_eval(xz.shape[0], xz.ctypes.data_as(C.POINTER(C.c_double)), 1e6, out.ctypes.data_as(C.POINTER(C.c_double)))
            dst = os.path.join(HERE, f"solcx_analytic_r{r}.npy")
            np.save(dst, np.concatenate([xz, out], axis=1))
            print("wrote", dst, out.shape)


if __name__ == "__main__":
    main()
