#!/bin/bash
# single-instance evaluation code (init pass): correctness subset, throughput, phase clocks, I-cache / LDS counters
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r02_call9}
mkdir -p $O
cd $ROOT
( time timeout 900 python -m pytest tests -m gpu -q -x -k "forcefield or bfgs or ff_ or etkdg or config_size or mmff or uff or embed" ) > $O/pytest.log 2>&1
timeout 300 python tools/bench_conformers.py --mols 2000 > $O/conf.json 2> $O/conf.err
NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols 400 > $O/phase.json 2> $O/phase.txt
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/tools/bench_conformers.py --mols 500"
for v in new old; do
  if [ $v = old ]; then export NVMOLKIT_AMD_LIB=$ROOT/tools/experiments/libnvmolkit_amd_ilp.so; fi
  timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES -f csv -d $O/pmc_icache_$v -- $BENCH > $O/pmc_icache_$v.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS -f csv -d $O/pmc_lds_$v -- $BENCH > $O/pmc_lds_$v.log 2>&1
done
python - "$O" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
res = {}
for sub in ("pmc_icache_new", "pmc_icache_old", "pmc_lds_new", "pmc_lds_old"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"{out}/{sub}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "bfgs_kernel" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
    res[sub] = {k: dict(v) for k, v in agg.items()}
json.dump(res, open(f"{out}/pmc_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cd $ROOT
tail -3 $O/pytest.log; cat $O/conf.json; grep "systems 4096\|systems 40[0-9][0-9]\|systems 39[0-9][0-9]" $O/phase.txt
