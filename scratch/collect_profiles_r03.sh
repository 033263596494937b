#!/bin/bash
# Round-3 profiles: kernel stats of the driver bench (LM block included), HBM traffic passes, SQ/LDS counters of both
# observation passes, MFMA-pipe counters of the dense Cholesky.  Run on the GPU box from the repo root: collect_profiles_r03.sh TAG
set -e -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench --output-format csv -- python3 $R/bench.py > $OUT/bench_under_rocprofv3.json 2> $OUT/bench.err
echo "stats done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-lm > $OUT/fetch.log 2>&1
echo "fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-lm > $OUT/write.log 2>&1
echo "write done"
timeout -s KILL 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $OUT/sq -o s --output-format csv -- python3 $R/scratch/ab_jtj.py 1000000 > $OUT/sq.log 2>&1
echo "sq done"
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES -d $OUT/mfma -o m --output-format csv -- python3 $R/scratch/chol_dense.py 30001 1 > $OUT/mfma.log 2>&1
echo "mfma done"
cd $R
python3 profiles/summarize_pmc.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
python3 scratch/pmcsum.py k_run_pass $OUT/sq > $OUT/pmc_run_pass_counters_10M.txt
python3 scratch/pmcsum.py k_ $OUT/mfma > $OUT/pmc_mfma_chol_n30001.txt
find $OUT -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/fetch -name '*counter_collection.csv' -exec cp {} $OUT/fetch_counter_collection.csv \;
find $OUT/write -name '*counter_collection.csv' -exec cp {} $OUT/write_counter_collection.csv \;
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq $OUT/mfma
tail -c 1500 $OUT/bench_under_rocprofv3.json; cat $OUT/pmc_run_pass_counters_10M.txt; grep -E "syrk|panel|potrf" $OUT/pmc_mfma_chol_n30001.txt; grep run_pass $OUT/pmc_traffic.txt; tail -2 $OUT/mfma.log
