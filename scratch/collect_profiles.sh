#!/bin/bash
# Round profiles of the driver bench: kernel stats, the two HBM traffic passes, one SQ pass.  Run on the GPU box from the repo root.
set -e -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench --output-format csv -- python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "stats done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $OUT/fetch.log 2>&1
echo "fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $OUT/write.log 2>&1
echo "write done"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/sq -o s --output-format csv -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $OUT/sq.log 2>&1
echo "sq done"
cd $R
python3 profiles/summarize_pmc.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
python3 scratch/pmcsum.py k_run_pass $OUT/sq > $OUT/sq_summary.txt
find $OUT -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/fetch -name '*counter_collection.csv' -exec cp {} $OUT/fetch_counter_collection.csv \;
find $OUT/write -name '*counter_collection.csv' -exec cp {} $OUT/write_counter_collection.csv \;
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq
cat $OUT/bench.json; cat $OUT/sq_summary.txt; grep run_pass $OUT/pmc_traffic.txt
