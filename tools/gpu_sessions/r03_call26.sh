#!/bin/bash
# Round 3, call 26: dense similarity with 16 384-row chunks (128-tile-tall supertiles, an XCD owns 64 x 16 tiles) against 8192-row chunks.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call26}
mkdir -p $O
cd $ROOT
( timeout 600 python -m pytest tests/test_similarity_gpu.py tests/test_full_size_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for c in 8192 16384 8192 16384; do
  timeout 300 python bench.py --chunk-rows $c --steps 3 --warmup 1 --cpu-seconds 0 --butina-n 0 --conformer-mols 0 --cfg1 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'chunk': $c, 'value': d['value'], 'frac': d['roofline']['frac'], 'avg_launch_ms': d['roofline']['avg_launch_ms']}))" | tee -a $O/chunk.jsonl
done
