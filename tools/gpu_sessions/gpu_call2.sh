#!/bin/bash
# round-2 GPU session 2: full GPU suite on the wave-per-row pass, LDS-policy comparison, profiles (conformers, Morgan, dense Butina), bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02_call2
mkdir -p $O
cd $ROOT
( time timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_config_size_gpu.py ) > $O/pytest.log 2>&1
( time timeout 600 python -m pytest tests/test_config_size_gpu.py -m gpu -q ) > $O/pytest_cfg.log 2>&1
for pol in 0 auto full; do
  NVMK_BFGS_LDS=$pol timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf_$pol.json 2> $O/conf_$pol.err
done
NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase_auto.json 2> $O/phase_auto.txt
timeout 600 bash tools/profile_conformers.sh r02_call2/prof 1000 auto > $O/prof.log 2>&1
for st in 32 64 128; do timeout 120 python tools/bench_morgan.py --mols 1000000 --stride $st >> $O/morgan.jsonl 2>> $O/morgan.err; done
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_morgan -- python $ROOT/tools/bench_morgan.py --mols 1000000 --stride 64 > $O/prof_morgan.log 2>&1 )
timeout 300 python tools/bench_butina_dense.py 20000 40000 60000 > $O/butina_dense.jsonl 2> $O/butina_dense.err
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_butina_dense -- python $ROOT/tools/bench_butina_dense.py 40000 > $O/prof_butina_dense.log 2>&1 )
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -4 $O/pytest.log; tail -4 $O/pytest_cfg.log; cat $O/conf_*.json; grep "bfgs profile" $O/phase_auto.txt | head -8; cat $O/morgan.jsonl $O/butina_dense.jsonl; tail -c 1200 $O/bench.json
