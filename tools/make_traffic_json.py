#!/usr/bin/env python3
"""Build profiles/<round>_traffic.json from the outputs of tools/pmc_traffic.sh (one file per refinement, named
..._r<refinement>.txt).  traffic_bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE reports half the bytes of
coalesced reads, profiles/r1_fetch_size_calibration.txt; WRITE_SIZE exact); counters are KiB per launch, averaged
over the launches of the pass, summed over the kernels of one A-block vmult."""
import ast
import json
import re
import sys

out = {"note": "HBM bytes per A-block vmult (all its kernels), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes "
               "(tools/pmc_traffic.sh via tools/round_profile.sh); traffic_bytes = 2 x FETCH_SIZE + WRITE_SIZE "
               "(gfx950 correction, profiles/r1_fetch_size_calibration.txt), traffic_bytes_lower = FETCH not doubled.",
       "convection_box_3d": {}}
for path in sys.argv[1:]:
    r = int(re.search(r"_r(\d+)\.txt$", path).group(1))
    kernels = {}
    for line in open(path):
        m = re.match(r"(.*?) (\{.*\})\s*$", line)
        if not m:
            continue
        name = m.group(1)
        if "k_brick_geo" in name:      # coefficient set-up, not part of a vmult
            continue
        short = "k_brick_fix" if "k_brick_fix" in name else "k_brickq" if "k_brick" in name else \
                "k_cells_blk" if "k_cells_blk" in name else "k_fix" if "k_fix" in name else "k_cells"
        d = ast.literal_eval(m.group(2))
        e = kernels.setdefault(short, {"FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0})
        e["FETCH_SIZE_KiB"] += d.get("FETCH_SIZE", 0.0)
        e["WRITE_SIZE_KiB"] += d.get("WRITE_SIZE", 0.0)
    f = sum(k["FETCH_SIZE_KiB"] for k in kernels.values()) * 1024
    w = sum(k["WRITE_SIZE_KiB"] for k in kernels.values()) * 1024
    out["convection_box_3d"][str(r)] = {"n_cells": 8 ** r, "fetch_size_bytes_raw": f, "write_size_bytes": w,
                                        "traffic_bytes": 2 * f + w, "traffic_bytes_lower": f + w, "kernels": kernels}
print(json.dumps(out, indent=1))
