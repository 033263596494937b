#!/bin/bash
# LDS / VALU counters of both passes on a regular and a random scene of the same size (scratch/regular_tiles.py): regular_counters.sh TAG SPEC...
set -e -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/summary.txt
for spec in "$@"; do
  timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $OUT/c -o c --output-format csv -- python3 $R/scratch/regular_tiles.py $spec > $OUT/run.log 2>&1
  echo "== $spec" >> $OUT/summary.txt
  python3 $R/scratch/pmcsum.py k_run_pass $OUT/c >> $OUT/summary.txt
  rm -rf $OUT/c
  echo "$spec done"
done
cat $OUT/summary.txt
