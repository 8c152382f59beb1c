#!/bin/bash
# Runs on the GPU box (via gpurun): L2-miss traffic (FETCH_SIZE, its own rocprofv3 --pmc pass) of the row-panel count kernel inside one
# fused Butina at 1M rows, for the product library and for every variant library beside it (nvmolkit_amd/lib/libnvmolkit_amd_<variant>.so).
# Output: gpurun_out/<session>/panel_fetch.json — KiB per launch as the counter gives them (the gfx950 doubling is applied by the reader).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:?output directory}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in "" $(ls $ROOT/nvmolkit_amd/lib/ | sed -n 's/^libnvmolkit_amd_\(.*\)\.so$/\1/p'); do
  NAME=${L:-product}
  NVMOLKIT_AMD_LIB=$ROOT/nvmolkit_amd/lib/libnvmolkit_amd${L:+_$L}.so timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/fetch_$NAME -- \
    python $ROOT/tools/bench_butina.py 1000000 > $OUT/fetch_$NAME.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = {}
for d in sorted(glob.glob(f"{sys.argv[1]}/fetch_*")):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[len("fetch_"):]
    per = {}
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "neighbor_count_panel_kernel" in k and r["Counter_Name"] == "FETCH_SIZE":
                form = "pair-emitting" if "<true" in k else "counting-only"
                per.setdefault(form, []).append(float(r["Counter_Value"]))
    out[name] = {form: {"launches": len(v), "FETCH_SIZE_KiB_per_launch": v} for form, v in per.items()}
    for line in open(f"{sys.argv[1]}/fetch_{name}.log"):
        if line.startswith("{"):
            out[name]["bench_butina_line_under_the_counter_pass"] = json.loads(line)
json.dump(out, open(f"{sys.argv[1]}/panel_fetch.json", "w"), indent=1)
print(json.dumps({k: {f: v for f, v in d.items() if f != "bench_butina_line_under_the_counter_pass"} for k, d in out.items()}, indent=1))
PY
