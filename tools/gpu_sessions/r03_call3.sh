#!/bin/bash
# Round 3, call 3: rest of the BFGS / ETKDG tests after the size-class change; concurrency of ETKDG batches; phase profile.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call3}
mkdir -p $O
cd $ROOT
( time timeout 900 python -m pytest tests/test_bfgs_parity_gpu.py tests/test_forcefield_gpu.py tests/test_etkdg_gpu.py tests/test_etkdg_driver_gpu.py tests/test_constraints.py tests/test_device_chain_gpu.py tests/test_config_size_gpu.py -m gpu -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for b in 1 2 3 4; do timeout 300 python tools/bench_conformers.py --mols 4000 --batches-per-gpu $b >> $O/conf4000_bpg.jsonl 2>> $O/conf.err; done
cat $O/conf4000_bpg.jsonl | cut -c1-400
NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase.json 2> $O/phase_profile.txt; grep profile $O/phase_profile.txt | sort | uniq -c | sort -rn | head -5;  grep "profile" $O/phase_profile.txt | head -40
