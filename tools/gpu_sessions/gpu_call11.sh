#!/bin/bash
# SMILES ingestion on the GPU, config-size tests, ingestion throughput, full bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r02_call11}
mkdir -p $O
cd $ROOT
( time timeout 900 python -m pytest tests/test_smiles_ingestion.py tests/test_config_size_gpu.py tests/test_morgan_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
timeout 300 python tools/bench_smiles_ingest.py --repeat 100 > $O/smiles_ingest.json 2> $O/smiles_ingest.err
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -8 $O/pytest.log; cat $O/smiles_ingest.json; tail -c 1800 $O/bench.json
