#!/bin/bash
# Round 3, call 9: hybrid BFGS with the wave thresholds (8 per CU up to 176 coordinates, 6 up to 232) and the new ETKDG batch defaults.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call9}
mkdir -p $O
cd $ROOT
( timeout 900 python -m pytest tests/test_bfgs_parity_gpu.py tests/test_forcefield_gpu.py tests/test_etkdg_gpu.py tests/test_etkdg_driver_gpu.py tests/test_constraints.py tests/test_device_chain_gpu.py tests/test_config_size_gpu.py tests/test_rmsd_gpu.py -m gpu -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf2000.json 2> $O/conf.err; cat $O/conf2000.json
timeout 300 python tools/bench_conformers.py --mols 10000 > $O/conf10000.json 2>> $O/conf.err; cat $O/conf10000.json
timeout 300 python tools/bench_conformers.py --mols 10000 --batch-size 4096 --batches-per-gpu 3 > $O/conf10000_4096x3.json 2>> $O/conf.err; cat $O/conf10000_4096x3.json
NVMK_BFGS_WAVE=0 timeout 300 python tools/bench_conformers.py --mols 10000 > $O/conf10000_nowave.json 2>> $O/conf.err; cat $O/conf10000_nowave.json
NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase.json 2> $O/phase_profile.txt; grep "profile" $O/phase_profile.txt | sort -t' ' -k7 -n -r | head -6
