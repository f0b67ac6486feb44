#!/bin/bash
# generic PMC passes over tools/prof_vmult.py: tools/pmc_passes.sh <outdir> <refinement> "<pass 1 counters>" "<pass 2 counters>" ...
out=$1; r=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pmc in "$@"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pmc --output-format csv -d $out/p$i -o pmc -- python3 tools/prof_vmult.py $r 5 box A > $out.p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out.p$i.log; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if "k_cells" in k or "k_fix" in k or "k_brick" in k:
        print(k)
        for c, v in sorted(d.items()):
            print(f"   {c:40s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
