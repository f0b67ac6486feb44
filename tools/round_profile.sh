#!/bin/bash
# Round-end measurement set, one gpurun call:  tools/round_profile.sh <tag>
#   1. plain bench line                           -> gpurun_out/<tag>/bench.json
#   2. the same command under rocprofv3 --stats   -> gpurun_out/<tag>/bench_kernel_stats.csv (+ bench_prof.json)
#   3. A-block vmult alone, r5 and r7             -> gpurun_out/<tag>/vmult_r{5,7}_kernel_stats.csv
#   4. FETCH_SIZE / WRITE_SIZE passes, r5 and r7  -> gpurun_out/<tag>/traffic.json (tools/make_traffic_json.py)
# Every GPU step has its own time limit; the first failure ends the script.
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
set -e
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -o bench -- python3 bench.py --steps 50 --warmup 5 > $out/bench_prof.json 2> $out/bench_prof.err
cp $(ls $out/prof_bench/*/*kernel_stats.csv $out/prof_bench/*kernel_stats.csv 2>/dev/null | head -1) $out/bench_kernel_stats.csv
echo "bench profile done"
for r in 5 7; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_v$r -o v -- python3 tools/prof_vmult.py $r 20 box A > $out/vmult_r$r.log 2>&1
  cp $(ls $out/prof_v$r/*/*kernel_stats.csv $out/prof_v$r/*kernel_stats.csv 2>/dev/null | head -1) $out/vmult_r${r}_kernel_stats.csv
  echo "vmult r$r profile done"
done
for r in 5 7; do
  bash tools/pmc_traffic.sh $out/pmc_r$r $r box > $out/pmc_r$r.txt
  echo "traffic r$r done"
done
python3 tools/make_traffic_json.py $out/pmc_r5.txt $out/pmc_r7.txt > $out/traffic.json
rm -rf $out/prof_bench $out/prof_v5 $out/prof_v7 $out/pmc_r5 $out/pmc_r7
ls -la $out
