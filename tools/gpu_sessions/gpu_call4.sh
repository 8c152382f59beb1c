#!/bin/bash
# H-pass microbenchmark matrix + parity + conformer throughput
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r02_call4}
mkdir -p $O
cd $ROOT
for n in 144 192; do for lds in 0 52 79; do for occ in 2 3; do
  timeout 60 tools/ubench_hess $n 4096 $lds $occ 50 >> $O/ubench.jsonl 2>> $O/ubench.err
done; done; done
( time timeout 600 python -m pytest tests/test_bfgs_parity_gpu.py tests/test_forcefield_gpu.py tests/test_etkdg_driver_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
for cfg in "auto 2" "auto 3" "0 3"; do
  set -- $cfg
  NVMK_BFGS_LDS=$1 NVMK_BFGS_OCC=$2 timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf_$1_occ$2.json 2> $O/conf_$1_occ$2.err
done
NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase_auto.json 2> $O/phase_auto.txt
cat $O/ubench.jsonl; tail -4 $O/pytest.log; for f in $O/conf_*.json; do echo $f; cut -c1-400 $f; done; grep "bfgs profile" $O/phase_auto.txt | sed -n '3,4p;8p'
