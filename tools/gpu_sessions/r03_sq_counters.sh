#!/bin/bash
# SQ counters of the BFGS kernels at the last kernel sources (one PMC pass, no tracing): VALU-active share of the wave cycles.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_sq}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS -f csv -d $O/pmc_sq -- \
  python $ROOT/tools/bench_conformers.py --mols 2000 > $O/pmc_sq.log 2>&1
cd $ROOT
python - "$O" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{out}/pmc_sq/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void nvmk::minim::", "")
        if "bfgs_kernel" in name:
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
res = {}
for name, c in sorted(agg.items()):
    w = c.get("SQ_WAVE_CYCLES", 0.0)
    res[name] = dict(c)
    if w:
        for k in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS"):
            res[name][k + "/SQ_WAVE_CYCLES"] = c.get(k, 0.0) / w
json.dump({"molecules": 2000, "command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS -- python tools/bench_conformers.py --mols 2000",
           "kernels": res}, open(f"{out}/sq_counters.json", "w"), indent=1)
for name, c in res.items():
    print(name, {k: round(v, 3) for k, v in c.items() if "/" in k})
PY
rm -rf $O/pmc_sq
