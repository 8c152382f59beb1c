#!/bin/bash
# A/B of the inverse-Hessian pass rework on one box: parity tests of the new library, then the conformer bench with the previous
# library (nvmolkit_amd/lib/libnvmolkit_amd_prev.so, built from the commit before) and the new one, alternating.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_hess_ab}
mkdir -p $O
cd $ROOT
( time timeout 600 python -m pytest tests/test_bfgs_parity_gpu.py tests/test_etkdg_gpu.py tests/test_forcefield_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for i in 1 2; do
  NVMOLKIT_AMD_LIB=$ROOT/nvmolkit_amd/lib/libnvmolkit_amd_prev.so timeout 300 python tools/bench_conformers.py --mols 10000 --repeat 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', {k: d[k] for k in d if k in ('mols_per_s_etkdg_plus_mmff','etkdg_s','mmff_s','etkdg_conformers','mmff_converged_frac')})" | tee -a $O/ab.txt
  timeout 300 python tools/bench_conformers.py --mols 10000 --repeat 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', {k: d[k] for k in d if k in ('mols_per_s_etkdg_plus_mmff','etkdg_s','mmff_s','etkdg_conformers','mmff_converged_frac')})" | tee -a $O/ab.txt
done
