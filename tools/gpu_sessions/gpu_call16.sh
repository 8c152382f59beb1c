#!/bin/bash
# end-of-round record: full GPU suite, HBM traffic of the dense launches, conformer kernel stats + PMC, full bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r02_call16}
mkdir -p $O
cd $ROOT
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
timeout 600 bash tools/profile_bench_traffic.sh > $O/traffic.log 2>&1
cp gpurun_out/pmc_traffic/pmc_hbm_traffic_bench_launch.json profiles/r02_similarity/pmc_hbm_traffic_bench_launch.json
timeout 600 bash tools/profile_conformers.sh ${1:-r02_call16}/prof 1000 auto > $O/prof.log 2>&1
NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase_auto.json 2> $O/phase_auto.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/trace_bench -- python $ROOT/bench.py --conformer-mols 2000 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $ROOT
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -4 $O/pytest.log; tail -c 1500 $O/bench.json
