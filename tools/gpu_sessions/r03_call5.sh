#!/bin/bash
# Round 3, call 5 (experiment): BFGS workgroups of 512 threads at 128 VGPRs (spilling build) — does more waves per SIMD pay?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call5}
mkdir -p $O
cd $ROOT
( timeout 600 python -m pytest tests/test_bfgs_parity_gpu.py -m gpu -q -x -k "trajectory_matches_oracle_for_every_system or bitwise" ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf2000.json 2> $O/conf2000.err; cat $O/conf2000.json
NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase.json 2> $O/phase_profile.txt; grep "profile" $O/phase_profile.txt | grep "systems 4\|systems 3" | head
