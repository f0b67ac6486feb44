#!/bin/bash
# HBM traffic counters only (FETCH_SIZE, WRITE_SIZE; separate passes) of the A-block vmult kernels
# usage: tools/pmc_traffic.sh <outdir> [refinement] [kind]     (AGMG_NUMBERING is passed through)
out=$1; r=${2:-7}; kind=${3:-box}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pmc --output-format csv -d $out/p$i -o pmc -- python3 tools/prof_vmult.py $r 5 $kind A > $out.p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out.p$i.log; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if ("k_cells" in k or "k_fix" in k or "k_brick" in k) and "k_brick_geo" not in k:   # (k_brick_geo: set-up, not part of a vmult)
        print(k, {c: sum(v) / len(v) for c, v in sorted(d.items())})
PY
