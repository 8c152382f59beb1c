#!/bin/bash
# Round 3, call 14: how large a system should one wave take?  (NVMK_BFGS_WAVE = largest coordinate count of the wave class)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call14}
mkdir -p $O
cd $ROOT
for w in 144 160 200 112; do
  NVMK_BFGS_WAVE=$w timeout 300 python tools/bench_conformers.py --mols 10000 --batch-size 16384 --batches-per-gpu 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'wave_max_n': $w, 'etkdg_s': d['etkdg_s'], 'mmff_s': d['mmff_s'], 'mols_per_s': d['mols_per_s_etkdg_plus_mmff']}))" | tee -a $O/wave_threshold.jsonl
done
