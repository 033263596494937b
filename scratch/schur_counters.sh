#!/bin/bash
# SQ / LDS counters of the Schur kernels (scratch/lm_bench.py): schur_counters.sh TAG
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $R/gpurun_out/$1 -o c --output-format csv -- python3 $R/scratch/lm_bench.py 2000 1000000 > $R/gpurun_out/$1.log 2>&1 || exit 1
python3 $R/scratch/pmcsum.py k_schur $R/gpurun_out/$1
rm -rf $R/gpurun_out/$1
