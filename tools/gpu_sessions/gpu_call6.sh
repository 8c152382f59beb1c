#!/bin/bash
# full GPU suite + bench + conformer profiles on the v4 pass
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r02_call6}
mkdir -p $O
cd $ROOT
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
timeout 600 bash tools/profile_conformers.sh ${1:-r02_call6}/prof 1000 auto > $O/prof.log 2>&1
for pol in 0 auto full; do NVMK_BFGS_LDS=$pol NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase_$pol.json 2> $O/phase_$pol.txt; done
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -5 $O/pytest.log; tail -c 2500 $O/bench.json
