#!/bin/bash
# Run on the GPU box (through gpurun): one bench run plus the rocprofv3 passes whose summaries are committed under
# profiles/.  usage: tools/profile_round.sh <tag> [bench args...]      -> gpurun_out/<tag>/...
# Passes (each its own run; PMC passes carry --kernel-trace only, as the pool requires):
#   1. plain bench (the JSON line)                      2. --kernel-trace --stats
#   3. --pmc SQ counters (two sets)                     4. --pmc FETCH_SIZE   5. --pmc WRITE_SIZE
set -u
TAG=${1:-prof}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 $REPO/bench.py --steps 20 --warmup 3 "$@" > $OUT/bench.json 2> $OUT/bench.err
B="$REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline $*"
# (the kernel trace runs long enough -- > 100 launches of the dominant kernel -- for its average to agree with the HIP events of the bench line)
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $REPO/bench.py --steps 12 --warmup 2 --no-cpu-baseline "$@" > $OUT/trace_bench.json 2> $OUT/trace.log
python3 $REPO/tools/rocprof_summary.py $(ls $OUT/trace/*/*.db $OUT/trace/*.db 2>/dev/null | head -1) $OUT/kernel_stats.txt > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq1 -o s -- python3 $B > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $OUT/sq2 -o s -- python3 $B > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $B > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $B > $OUT/write.log 2>&1
F=$(ls $OUT/fetch/*/*counter_collection.csv $OUT/fetch/*counter_collection.csv 2>/dev/null | head -1)
W=$(ls $OUT/write/*/*counter_collection.csv $OUT/write/*counter_collection.csv 2>/dev/null | head -1)
python3 $REPO/tools/pmc_traffic.py $F $W 8 $OUT/pmc_hbm_traffic.json $OUT/traffic.json > $OUT/pmc_hbm_traffic.txt 2>&1
for s in sq1 sq2; do
  C=$(ls $OUT/$s/*/*counter_collection.csv $OUT/$s/*counter_collection.csv 2>/dev/null | head -1)
  python3 $REPO/tools/pmc_sq_summary.py $C > $OUT/pmc_$s.txt 2>&1
done
# keep what travels back small: drop the raw traces (csv of the PMC passes are a few MB; the sqlite db is kept)
find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh $OUT | tail -1
cat $OUT/bench.json | head -c 3000; echo; head -20 $OUT/kernel_stats.txt; cat $OUT/pmc_sq1.txt $OUT/pmc_sq2.txt; cat $OUT/pmc_hbm_traffic.txt
