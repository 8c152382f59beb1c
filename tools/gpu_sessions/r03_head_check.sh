#!/bin/bash
# Head of the round: smoke(), the clustering tests (the last product change) and the default bench line.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_head_check}
mkdir -p $O
cd $ROOT
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
( timeout 300 python -m pytest tests/test_clustering_gpu.py -m gpu -q -x 2>&1 | tail -2 ) | tee $O/pytest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-200 $O/bench.json
