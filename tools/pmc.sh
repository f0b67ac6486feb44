#!/bin/bash
# PMC passes for the A-block vmult kernels (one rocprofv3 run per pass; see MI355X_MICROARCH.md)
# usage: tools/pmc.sh <outdir> [refinement] [kind]
out=$1; r=${2:-7}; kind=${3:-box}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
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
    if "k_cells" in k or "k_fix" in k or "k_pull" in k or "k_brick" in k:
        print(k)
        for c, v in sorted(d.items()):
            print(f"   {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
