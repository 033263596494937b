#!/bin/bash
# Kernel stats of the LM micro-benchmark (Cholesky kernels): chol_trace.sh TAG [K N]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$1 -o lm --output-format csv -- python3 $R/scratch/lm_bench.py ${2:-2000} ${3:-1000000} > $R/gpurun_out/$1.log 2>&1 || exit 1
cd $R
f=$(find gpurun_out/$1 -name "*kernel_stats.csv")
cut -c1-150 $f | grep -E "panel|potrf|trsm|syrk|solve|schur|diag|Name"
find gpurun_out/$1 -name "*.csv" ! -name "*kernel_stats.csv" -delete
grep -E "chol|residual" gpurun_out/$1.log
