#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the N x N neighbour-count pass
# (tools/bench_butina.py --skip-butina).  Outputs under gpurun_out/${PROF_NAME:-prof_counts}/ ; copy summaries into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${PROF_NAME:-prof_counts}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
N=${1:-262144}
BENCH="python $ROOT/tools/bench_butina.py $N --skip-butina"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS -f csv -d $OUT/pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -f csv -d $OUT/pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD -f csv -d $OUT/pmc3 -- $BENCH > $OUT/pmc3.log 2>&1
echo done
