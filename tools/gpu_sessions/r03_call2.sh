#!/bin/bash
# Round 3, call 2: the size-class BFGS (classes A / B / C, persistent large classes on side streams).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call2}
mkdir -p $O
cd $ROOT
( time timeout 900 python -m pytest tests/test_bfgs_parity_gpu.py tests/test_forcefield_gpu.py tests/test_etkdg_gpu.py tests/test_constraints.py tests/test_device_chain_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf2000.json 2> $O/conf2000.err; cat $O/conf2000.json
timeout 600 python tools/bench_mixed_sizes.py --mols 2000 > $O/mixed.jsonl 2> $O/mixed.err; cat $O/mixed.jsonl; tail -3 $O/mixed.err
