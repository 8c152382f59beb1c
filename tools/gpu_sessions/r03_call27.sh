#!/bin/bash
# Round 3, call 27: count kernel with two waves of 64 x 128 per tile against four waves of 64 x 64.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call27}
mkdir -p $O
cd $ROOT
( NVMK_COUNT_WAVES=2 timeout 600 python -m pytest tests/test_clustering_gpu.py tests/test_benchmark_molecules_gpu.py -m gpu -q -x -k "not valu" ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for w in 4 2 4 2; do
  NVMK_COUNT_WAVES=$w timeout 300 python tools/bench_butina.py 1000000 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'waves': $w, 'neighbor_counts_s': d['neighbor_counts_s'], 'fused_butina_s': d['fused_butina_s']}))" | tee -a $O/count_waves.jsonl
done
