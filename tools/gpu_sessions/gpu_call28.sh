# final record of the round: traffic file for the final dense kernel, pure-store rate of the box, full bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_call28; mkdir -p $O
bash tools/profile_bench_traffic.sh > $O/traffic.log 2>&1
cp gpurun_out/pmc_traffic/pmc_hbm_traffic_bench_launch.json profiles/r02_similarity/pmc_hbm_traffic_bench_launch.json
tools/ubench_store > $O/ubench_store.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/trace_bench -- python $GRAFT_REPO_ROOT/bench.py --conformer-mols 2000 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err
cd $GRAFT_REPO_ROOT
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
head -3 $O/ubench_store.txt; tail -c 600 $O/bench.json
