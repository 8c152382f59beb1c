#!/bin/bash
# prefetched term batches + diagonal pair order: suite, A/B throughput and phase clocks
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r02_call7}
mkdir -p $O
cd $ROOT
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1
for ord in input diagonal; do
  NVMK_PAIR_ORDER=$ord timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf_$ord.json 2> $O/conf_$ord.err
  NVMK_PAIR_ORDER=$ord NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase_$ord.json 2> $O/phase_$ord.txt
done
NVMK_BFGS_OCC=3 timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf_diagonal_occ3.json 2> $O/conf_diagonal_occ3.err
tail -5 $O/pytest.log; cat $O/conf_*.json; grep "systems 4096\|systems 40[0-9][0-9]\|systems 39[0-9][0-9]" $O/phase_*.txt
