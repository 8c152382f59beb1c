#!/bin/bash
# First GPU call of the NEXT round (round 4): the state round 3 ended in, measured again before anything changes — the whole GPU
# suite, smoke(), the default bench line, its rocprofv3 summary and both PMC traffic files — and the two things round 3 wrote
# after its last full session that have only run in part on a GPU: the ingestion benchmark with the corrected thread sweep and
# the reference's benchmark shapes.
#   gpurun --timeout 1800 -- 'bash tools/gpu_sessions/next_first_call.sh r04_call1'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=${1:-r04_call1}
O=$ROOT/gpurun_out/$NAME
mkdir -p $O
cd $ROOT
bash tools/gpu_sessions/r03_final.sh $NAME
timeout 200 python tools/bench_smiles_ingest.py --repeat 100 > $O/smiles_ingest.json 2> $O/smiles_ingest.err
timeout 120 python tools/bench_smiles_threads.py > $O/smiles_threads.json 2> $O/smiles_threads.err
timeout 400 python tools/reference_benchmark_shapes.py --runs 3 > $O/reference_shapes.jsonl 2> $O/reference_shapes.err
cat $O/smiles_ingest.json; cat $O/smiles_threads.json; tail -5 $O/reference_shapes.jsonl
