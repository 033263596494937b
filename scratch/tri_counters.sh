#!/bin/bash
# issue counters of k_triangulate at cfg3: tri_counters.sh TAG [BA_LIB]
R=$GRAFT_REPO_ROOT
export BA_LIB=$2
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/$1 -o c --output-format csv -- python3 $R/scratch/tri_bench.py > $R/gpurun_out/$1.log 2>&1 || exit 1
python3 $R/scratch/pmcsum.py k_triangulate $R/gpurun_out/$1 > $R/gpurun_out/$1.txt
rm -rf $R/gpurun_out/$1
cat $R/gpurun_out/$1.txt
