#!/bin/bash
# Fused Butina after the host-side canonicalisation went parallel: every clustering test, then the 1M-row block of bench.py.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_butina_host}
mkdir -p $O
cd $ROOT
( timeout 500 python -m pytest tests/test_clustering_gpu.py tests/test_full_size_gpu.py -m gpu -q -x 2>&1 | tail -3 ) | tee $O/pytest.txt
timeout 300 python bench.py --steps 1 --warmup 0 --cpu-seconds 0 --cfg1 0 --conformer-mols 0 > $O/bench.json 2> $O/bench.err
python - <<'P'
import json,sys
d=json.loads(open(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/r03_butina_host/bench.json').read().strip().splitlines()[-1])
b=d['secondary']['fused_butina']; print('butina seconds', b['seconds'], 'clusters', b['n_clusters'])
P
