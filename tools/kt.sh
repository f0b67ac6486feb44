#!/bin/bash
# usage: kt.sh <lib> <r> <n> ; prints kernel average durations of k_brickq and k_brick_fix
cd /tmp && export TMPDIR=/tmp
L=$1; R=$2; N=$3
rm -rf /tmp/kt_$$
AGMG_LIB=$GRAFT_REPO_ROOT/$L AGMG_BRICKS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$$ -o kt -- python3 $GRAFT_REPO_ROOT/tools/prof_vmult.py $R $N box A > /dev/null 2>&1
grep -E "k_brickq|k_brick_fix" /tmp/kt_$$/kt_kernel_stats.csv | awk -F'","' '{n=split($1,a,"::"); printf "%-60s calls %s avg_ns %s\n", substr(a[n],1,58), $2, $4}'
