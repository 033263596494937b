#!/bin/bash
# kernel stats of the "next" rows at cfg3 size: next_rows_trace.sh TAG
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$1_a -o a --output-format csv -- python3 $R/scratch/init_bench.py > $R/gpurun_out/$1_init.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$1_b -o b --output-format csv -- python3 $R/scratch/pairs_bench.py > $R/gpurun_out/$1_pairs.log 2>&1 || exit 1
cd $R
for d in $1_a $1_b; do f=$(find gpurun_out/$d -name "*kernel_stats.csv"); cp $f gpurun_out/$d.kernel_stats.csv; rm -rf gpurun_out/$d; done
tail -2 gpurun_out/$1_init.log; tail -2 gpurun_out/$1_pairs.log
