#!/bin/bash
# Round 3, call 17: one-wave H pass with rows 0..63 two at a time.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call23}
mkdir -p $O
cd $ROOT
( timeout 900 python -m pytest tests/test_bfgs_parity_gpu.py tests/test_forcefield_gpu.py tests/test_etkdg_gpu.py tests/test_constraints.py -m gpu -q ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 300 python tools/bench_conformers.py --mols 10000 > $O/conf10000.json 2> $O/conf.err; cat $O/conf10000.json
NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase.json 2> $O/phase_profile.txt; grep "profile" $O/phase_profile.txt | sort -t' ' -k7 -n -r | head -5
