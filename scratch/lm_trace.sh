#!/bin/bash
# Kernel stats of whole LM iterations: lm_trace.sh TAG K N ITERS
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$1 -o lm --output-format csv -- python3 $R/scratch/lm_loop.py ${2:-2000} ${3:-1000000} ${4:-6} > $R/gpurun_out/$1.log 2>&1 || exit 1
cd $R
f=$(find gpurun_out/$1 -name "*kernel_stats.csv")
cut -c1-130 $f | head -24
find gpurun_out/$1 -name "*.csv" ! -name "*kernel_stats.csv" -delete
tail -1 gpurun_out/$1.log
