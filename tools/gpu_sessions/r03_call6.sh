#!/bin/bash
# Round 3, call 6: LDS bins of the BFGS launcher (default 256-thread build) + the experiment of one WAVE per system (64-thread build).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call6}
mkdir -p $O
cd $ROOT
( timeout 600 python -m pytest tests/test_bfgs_parity_gpu.py tests/test_forcefield_gpu.py tests/test_etkdg_gpu.py -m gpu -q -x ) > $O/pytest256.log 2>&1
tail -3 $O/pytest256.log
timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf2000_t256.json 2> $O/conf2000.err; cat $O/conf2000_t256.json
export NVMOLKIT_AMD_LIB=$ROOT/nvmolkit_amd/lib/libnvmolkit_amd_t64.so
( timeout 600 python -m pytest tests/test_bfgs_parity_gpu.py -m gpu -q -x -k "trajectory_matches_oracle_for_every_system or bitwise" ) > $O/pytest64.log 2>&1
tail -3 $O/pytest64.log
timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf2000_t64.json 2> $O/conf2000_t64.err; cat $O/conf2000_t64.json
NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase64.json 2> $O/phase_profile64.txt; grep "profile" $O/phase_profile64.txt | sort -t' ' -k7 -n -r | head -12
