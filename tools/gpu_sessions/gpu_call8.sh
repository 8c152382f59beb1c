#!/bin/bash
# run_radial (4 pair terms side by side): default scheduler vs max-ilp scheduler; correctness subset first
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r02_call8}
mkdir -p $O
cd $ROOT
( time timeout 900 python -m pytest tests -m gpu -q -x -k "forcefield or bfgs or ff_ or etkdg or config_size or mmff or uff or embed" ) > $O/pytest.log 2>&1
for v in default ilp; do
  if [ $v = ilp ]; then export NVMOLKIT_AMD_LIB=$ROOT/tools/experiments/libnvmolkit_amd_ilp.so; fi
  timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf_$v.json 2> $O/conf_$v.err
  NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase_$v.json 2> $O/phase_$v.txt
done
tail -5 $O/pytest.log; cat $O/conf_*.json; grep "systems 4096\|systems 40[0-9][0-9]\|systems 39[0-9][0-9]" $O/phase_*.txt
