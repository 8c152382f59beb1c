#!/bin/bash
# Runs on the GPU box (via gpurun): VALU microbenchmark + rocprofv3 kernel trace / PMC passes of the
# dense similarity bench.  Outputs under gpurun_out/prof_dense/ ; copy the summaries into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${PROF_NAME:-prof_dense}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
N=${1:-65536}
BENCH="python $ROOT/bench.py --path ${BENCH_PATH:-mfma} --n-query $N --n-ref $N --chunk-rows $N --steps 2 --warmup 1 --cpu-seconds 0"
rocprofv3 -L > $OUT/counters_list.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS -f csv -d $OUT/pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -f csv -d $OUT/pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc3 -- $BENCH > $OUT/pmc3.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc4 -- $BENCH > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
echo done
