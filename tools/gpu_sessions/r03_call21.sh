#!/bin/bash
# Round 3, call 21: a two-wave class between the one-wave and the four-wave kernels?  (NVMK_BFGS_WAVE2 = its largest system)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call21}
mkdir -p $O
cd $ROOT
( NVMK_BFGS_WAVE2=320 timeout 600 python -m pytest tests/test_bfgs_parity_gpu.py -m gpu -q -x -k "trajectory_matches_oracle_for_every_system or bitwise or mixed or restarts" ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for w in 0 240 288 336 400; do
  NVMK_BFGS_WAVE2=$w timeout 300 python tools/bench_conformers.py --mols 10000 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'wave2_max_n': $w, 'etkdg_s': d['etkdg_s'], 'mmff_s': d['mmff_s'], 'mols_per_s': d['mols_per_s_etkdg_plus_mmff']}))" | tee -a $O/wave2_threshold.jsonl
done
