#!/bin/bash
# LDS array cycles and bank-conflict cycles of both passes for experimental builds: lds_conflicts.sh NAME...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  echo "== $v"
  export BA_LIB=$R/scratch/libs/$v.so
  timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES -d $R/gpurun_out/ldsc_$v -o c --output-format csv -- python3 $R/scratch/ab_jtj.py 1000000 > $R/gpurun_out/ldsc_$v.log 2>&1 || exit 1
  python3 $R/scratch/pmcsum.py k_run_pass $R/gpurun_out/ldsc_$v | awk '{print $3, $4, $5, $6}' | sed 's/ba::RecI16>(ba::RunArgs)//'
  grep jtj $R/gpurun_out/ldsc_$v.log
  rm -rf $R/gpurun_out/ldsc_$v
done
