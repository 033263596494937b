#!/bin/bash
# Counters of the stand-alone J^T J and gradient passes (scratch/ab_jtj.py) at one shard size: jtj_counters.sh TAG POINTS
set -e -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
N=$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" -d $OUT/$name -o c --output-format csv -- python3 $R/scratch/ab_jtj.py $N > $OUT/$name.log 2>&1
  python3 $R/scratch/pmcsum.py k_run_pass $OUT/$name >> $OUT/summary_$N.txt
  rm -rf $OUT/$name
  echo "$name done"
}
: > $OUT/summary_$N.txt
run sq1 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
run sq2 SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
cat $OUT/summary_$N.txt
