#!/bin/bash
# Round 3, call 12: large classes placed before the small one (host waits for their workgroups to start).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call12}
mkdir -p $O
cd $ROOT
( timeout 900 python -m pytest tests/test_bfgs_parity_gpu.py tests/test_etkdg_gpu.py tests/test_device_chain_gpu.py -m gpu -q ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf2000.json 2> $O/conf.err; cat $O/conf2000.json
timeout 300 python tools/bench_conformers.py --mols 10000 > $O/conf10000.json 2>> $O/conf.err; cat $O/conf10000.json
timeout 300 python tools/bench_conformers.py --mols 10000 --batch-size 4096 --batches-per-gpu 3 > $O/conf10000_4096x3.json 2>> $O/conf.err; cat $O/conf10000_4096x3.json
timeout 300 python tools/bench_conformers.py --mols 10000 --batch-size 16384 --batches-per-gpu 1 > $O/conf10000_16384x1.json 2>> $O/conf.err; cat $O/conf10000_16384x1.json
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/trace -- python $ROOT/tools/bench_conformers.py --mols 2000 > $O/trace.log 2>&1
python - $O/trace <<'P'
import csv, glob, sys, collections
for f in glob.glob(f"{sys.argv[1]}/**/*_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6)
P
