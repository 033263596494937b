#!/bin/bash
# kernel time of k_resect for library variants: resect_abl.sh TAG LIB...
R=$GRAFT_REPO_ROOT
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export BA_LIB=$R/scratch/libs/$v.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_$v -o a --output-format csv -- python3 $R/scratch/init_bench.py > $R/gpurun_out/${tag}_$v.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/${tag}_$v -name "*kernel_stats.csv"); echo $v; grep k_resect $f | cut -c1-90; rm -rf $R/gpurun_out/${tag}_$v
done
