#!/bin/bash
# Round 3, final GPU session: PMC traffic of the dense launches and of the conformer kernels (separate --pmc passes), then — with
# those files in place — the full GPU suite, smoke, the default bench line, its rocprofv3 --kernel-trace --stats summary, the
# mixed-size conformer batches and the pure-store ceiling of the same box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_final}
mkdir -p $O
cd $ROOT
bash tools/profile_bench_traffic.sh > $O/pmc_dense.log 2>&1
cp gpurun_out/pmc_traffic/pmc_hbm_traffic_bench_launch.json $O/ 2>/dev/null
mkdir -p profiles/r03_similarity profiles/r03_conformers
cp gpurun_out/pmc_traffic/pmc_hbm_traffic_bench_launch.json profiles/r03_similarity/ 2>/dev/null
bash tools/profile_conformer_traffic.sh 2000 > $O/pmc_conformers.log 2>&1
cp gpurun_out/pmc_traffic/pmc_hbm_traffic_conformers.json $O/ 2>/dev/null
cp gpurun_out/pmc_traffic/pmc_hbm_traffic_conformers.json profiles/r03_conformers/ 2>/dev/null
rm -rf gpurun_out/pmc_traffic/fetch gpurun_out/pmc_traffic/write gpurun_out/pmc_traffic/conf_fetch gpurun_out/pmc_traffic/conf_write
cd $ROOT
( time timeout 1100 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_bench -- python $ROOT/bench.py > $O/bench_profiled.json 2> $O/bench_profiled.err
cd $ROOT
find $O/prof_bench -name "*_kernel_stats.csv" -exec cp {} $O/kernel_stats_bench_py.csv \;
find $O/prof_bench -name "*_kernel_trace.csv" -delete
head -8 $O/kernel_stats_bench_py.csv | cut -c1-200
timeout 600 python tools/bench_mixed_sizes.py --mols 2000 > $O/mixed_sizes.jsonl 2> $O/mixed.err; cat $O/mixed_sizes.jsonl | cut -c1-300
hipcc --offload-arch=gfx950 -O3 tools/ubench_store.hip -o /tmp/ubench_store 2>/dev/null && /tmp/ubench_store > $O/ubench_store_same_box.txt 2>&1; cat $O/ubench_store_same_box.txt
