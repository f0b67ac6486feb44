#!/usr/bin/env python3
"""Writes the 75 sinker centres of the NSinker benchmarks (benchmarks/nsinker/nsinker.cc:203-277 =
benchmarks/nsinker_spherical_shell/nsinker.cc:204-278: a table of numbers) to nsinker_centers.npy.
Run in the build container: python aspect_amd/data/make_nsinker_centers.py"""
import os
import re
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
txt = open(os.path.join(REF, "benchmarks", "nsinker_spherical_shell", "nsinker.cc")).read()
c = np.zeros((75, 3))
seen = set()
for m in re.finditer(r"centers\[(\d+)\]\s*=\s*Point<3>\(([^,]+),([^,]+),([^)]+)\)", txt):
    i = int(m.group(1))
    c[i] = [float(m.group(k)) for k in (2, 3, 4)]
    seen.add(i)
assert seen == set(range(75))
box = open(os.path.join(REF, "benchmarks", "nsinker", "nsinker.cc")).read()
for m in re.finditer(r"centers\[(\d+)\]\s*=\s*Point<3>\(([^,]+),([^,]+),([^)]+)\)", box):
    assert np.allclose(c[int(m.group(1))], [float(m.group(k)) for k in (2, 3, 4)])
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "nsinker_centers.npy"), c)
print("wrote 75 centres")
