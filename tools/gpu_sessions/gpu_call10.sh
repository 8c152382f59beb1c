#!/bin/bash
# groups rotated across the waves: correctness subset, throughput, phase clocks
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r02_call10}
mkdir -p $O
cd $ROOT
( time timeout 900 python -m pytest tests -m gpu -q -x -k "forcefield or bfgs or ff_ or etkdg or config_size or mmff or uff or embed" ) > $O/pytest.log 2>&1
timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf.json 2> $O/conf.err
NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase.json 2> $O/phase.txt
tail -3 $O/pytest.log; cat $O/conf.json; grep "systems 4096\|systems 40[0-9][0-9]\|systems 39[0-9][0-9]" $O/phase.txt
