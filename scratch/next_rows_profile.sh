#!/bin/bash
# kernel stats of the "next" rows at cfg3 size for profiles/: next_rows_profile.sh TAG
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for s in tri_bench init_bench pairs_bench; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$1_$s -o a --output-format csv -- python3 $R/scratch/$s.py > $R/gpurun_out/$1_$s.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/$1_$s -name "*kernel_stats.csv"); cp $f $R/gpurun_out/$1_$s.kernel_stats.csv; rm -rf $R/gpurun_out/$1_$s
done
grep -h "k_triangulate\|k_resect\|k_pair_homography" $R/gpurun_out/$1_*.kernel_stats.csv | cut -c1-120
