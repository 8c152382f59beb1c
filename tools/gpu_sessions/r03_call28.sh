#!/bin/bash
# Round 3, call 28: XCD group size of the BFGS hand-out with the one- / two-wave classes (default 16).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call28}
mkdir -p $O
cd $ROOT
for g in 16 8 32 64 1; do
  NVMK_BFGS_XCD_GROUP=$g timeout 300 python tools/bench_conformers.py --mols 10000 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'xcd_group': $g, 'etkdg_s': d['etkdg_s'], 'mmff_s': d['mmff_s'], 'mols_per_s': d['mols_per_s_etkdg_plus_mmff']}))" | tee -a $O/xcd_group.jsonl
done
