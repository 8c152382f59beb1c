#!/bin/bash
# Eight HBM rows per group in the one-chunk ranges of the inverse-Hessian pass: parity tests, A/B against the previous library
# on one box, then the PMC traffic passes of the new kernels (kept only if the change is).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_rows8_ab}
mkdir -p $O
cd $ROOT
( timeout 300 python -m pytest tests/test_bfgs_parity_gpu.py tests/test_etkdg_gpu.py tests/test_forcefield_gpu.py -m gpu -q -x 2>&1 | tail -2 ) | tee $O/pytest.txt
for i in 1 2; do
  NVMOLKIT_AMD_LIB=$ROOT/nvmolkit_amd/lib/libnvmolkit_amd_prev.so timeout 200 python tools/bench_conformers.py --mols 10000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['etkdg_s'], d['mmff_s'], d['mols_per_s_etkdg_plus_mmff'], d['etkdg_conformers'], d['mmff_converged_frac'])" | tee -a $O/ab.txt
  timeout 200 python tools/bench_conformers.py --mols 10000 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['etkdg_s'], d['mmff_s'], d['mols_per_s_etkdg_plus_mmff'], d['etkdg_conformers'], d['mmff_converged_frac'])" | tee -a $O/ab.txt
done
bash tools/profile_conformer_traffic.sh 2000 > $O/pmc_conformers.log 2>&1
cp gpurun_out/pmc_traffic/pmc_hbm_traffic_conformers.json $O/ 2>/dev/null
rm -rf gpurun_out/pmc_traffic/conf_fetch gpurun_out/pmc_traffic/conf_write
tail -3 $O/pmc_conformers.log | cut -c1-200
