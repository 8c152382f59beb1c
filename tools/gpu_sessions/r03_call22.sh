#!/bin/bash
# Round 3, call 22: one-wave threshold again with the two-wave class behind it.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call22}
mkdir -p $O
cd $ROOT
for pair in "144 256" "160 256" "176 256" "192 256" "176 208" "128 232"; do
  set -- $pair
  NVMK_BFGS_WAVE=$1 NVMK_BFGS_WAVE2=$2 timeout 300 python tools/bench_conformers.py --mols 10000 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'wave_max_n': $1, 'wave2_max_n': $2, 'etkdg_s': d['etkdg_s'], 'mmff_s': d['mmff_s'], 'mols_per_s': d['mols_per_s_etkdg_plus_mmff']}))" | tee -a $O/wave12_threshold.jsonl
done
