#!/bin/bash
# round-2 GPU session 1: regression + new parity tests, LDS-policy comparison, phase timers, profiles, full bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02_call1
mkdir -p $O
cd $ROOT
python -c "import rdkit" > $O/rdkit_probe.log 2>&1; nproc > $O/nproc.log
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
for pol in 0 auto full; do
  NVMK_BFGS_LDS=$pol timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf_$pol.json 2> $O/conf_$pol.err
done
for pol in 0 auto full; do
  NVMK_BFGS_LDS=$pol NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase_$pol.json 2> $O/phase_$pol.txt
done
timeout 900 bash tools/profile_conformers.sh r02_call1/prof 1000 0 auto > $O/prof.log 2>&1
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest.log; cat $O/conf_*.json; grep "bfgs profile" $O/phase_auto.txt | head -8; tail -c 1500 $O/bench.json
