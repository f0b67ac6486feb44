#!/bin/bash
# usage: tools/calib.sh <outdir>   (two PMC passes over tools/calib_fetch)
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -o pmc -- ./tools/calib_fetch > $out.f.log 2>&1 || { echo "fetch pass failed"; tail -3 $out.f.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -o pmc -- ./tools/calib_fetch > $out.w.log 2>&1 || { echo "write pass failed"; tail -3 $out.w.log; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, {c: sum(v) / len(v) for c, v in d.items()})
PY
