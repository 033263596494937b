#!/bin/bash
# matrix-core and issue counters of the Schur kernel in LM iterations at cfg3: schur_counters2.sh TAG
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/$1 -o c --output-format csv -- python3 $R/scratch/lm_loop.py 2000 1000000 6 > $R/gpurun_out/$1.log 2>&1 || exit 1
python3 $R/scratch/pmcsum.py k_schur_gemm $R/gpurun_out/$1 > $R/gpurun_out/$1.txt
rm -rf $R/gpurun_out/$1
cat $R/gpurun_out/$1.txt
