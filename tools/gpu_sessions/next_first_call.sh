#!/bin/bash
# First GPU call of the next round (tests/test_zz_gpu_checks_added_late.py holds the GPU tests that have never run on a GPU): everything that was built after round 2's GPU budget ran out gets its first run on an
# MI355X here — the full GPU suite (the SMILES staging path, the GH-84 regression through SMILES, the re-worded
# BatchedForcefield error are new), bench.py, the reworked ingestion end to end and the reference's own benchmark shapes.
#   gpurun --timeout 1500 -- 'bash tools/gpu_sessions/next_first_call.sh r03_call1'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call1}
mkdir -p $O
cd $ROOT
( time timeout 1100 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
timeout 200 python tools/bench_smiles_ingest.py --repeat 100 > $O/smiles_ingest.json 2> $O/smiles_ingest.err
timeout 400 python tools/reference_benchmark_shapes.py --runs 3 > $O/reference_shapes.jsonl 2> $O/reference_shapes.err
tail -3 $O/pytest.log; cat $O/smoke.log | tail -1; cut -c1-600 $O/bench.json; cat $O/smiles_ingest.json; tail -5 $O/reference_shapes.jsonl
