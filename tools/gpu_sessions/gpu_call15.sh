#!/bin/bash
# dense kernel with interleaved B columns (no lane exchange in the epilogue): parity + bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r02_call15}
mkdir -p $O
cd $ROOT
( time timeout 900 python -m pytest tests/test_similarity_gpu.py tests/test_full_size_gpu.py tests/test_clustering_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
for i in 1 2; do timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --butina-n 0 --conformer-mols 0 --cfg1 0 > $O/bench_$i.json 2> $O/bench_$i.err; done
tail -4 $O/pytest.log; cat $O/bench_1.json | head -c 600; echo; cat $O/bench_2.json | head -c 300
