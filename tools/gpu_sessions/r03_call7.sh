#!/bin/bash
# Round 3, call 8: hybrid BFGS (one wave per system up to 232 coordinates, four waves beyond): parity, throughput, batch sweep, A/B.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call8}
mkdir -p $O
cd $ROOT
( timeout 900 python -m pytest tests/test_bfgs_parity_gpu.py tests/test_forcefield_gpu.py tests/test_etkdg_gpu.py tests/test_etkdg_driver_gpu.py tests/test_constraints.py tests/test_device_chain_gpu.py tests/test_config_size_gpu.py -m gpu -q ) > $O/pytest.log 2>&1
tail -15 $O/pytest.log
timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf2000.json 2> $O/conf2000.err; cat $O/conf2000.json
NVMK_BFGS_WAVE=0 timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf2000_nowave.json 2>> $O/conf2000.err; cat $O/conf2000_nowave.json
timeout 600 python tools/sweep_embed_batch.py --mols 10000 > $O/sweep.jsonl 2> $O/sweep.err; cat $O/sweep.jsonl
NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase.json 2> $O/phase_profile.txt; grep "profile" $O/phase_profile.txt | sort -t' ' -k7 -n -r | head -12
