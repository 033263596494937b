#!/bin/bash
# Timeline of one LM iteration: lm_trace2.sh TAG K N ITERS
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/$1 -o lm --output-format csv -- python3 $R/scratch/lm_loop.py ${2:-2000} ${3:-1000000} ${4:-6} > $R/gpurun_out/$1.log 2>&1 || exit 1
cd $R
f=$(find gpurun_out/$1 -name "*kernel_trace.csv")
python3 scratch/lm_timeline.py $f > gpurun_out/$1_timeline.txt
python3 scratch/lm_gaps.py $f >> gpurun_out/$1_timeline.txt
rm -rf gpurun_out/$1
tail -1 gpurun_out/$1.log
