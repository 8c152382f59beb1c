#!/bin/bash
# round-2 GPU session 3: BFGS parity on the reworked pass + hand-derived gradients, occupancy / LDS policy comparison
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02_call3
mkdir -p $O
cd $ROOT
( time timeout 600 python -m pytest tests/test_bfgs_parity_gpu.py tests/test_forcefield_gpu.py tests/test_constraints.py tests/test_etkdg_gpu.py tests/test_device_chain_gpu.py tests/test_etkdg_driver_gpu.py tests/test_threads_gpu.py -m gpu -q ) > $O/pytest.log 2>&1
for cfg in "auto 2" "auto 3" "0 2" "0 3" "full 2"; do
  set -- $cfg
  NVMK_BFGS_LDS=$1 NVMK_BFGS_OCC=$2 timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf_$1_occ$2.json 2> $O/conf_$1_occ$2.err
done
NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase_auto.json 2> $O/phase_auto.txt
tail -5 $O/pytest.log; for f in $O/conf_*.json; do echo $f; cat $f; done; grep "bfgs profile" $O/phase_auto.txt | head -8
