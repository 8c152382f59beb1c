#!/bin/bash
# Runs on the GPU box (via gpurun): HBM traffic of the bench's own dense launches, FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc passes (never combined with tracing).  Output: gpurun_out/pmc_traffic/pmc_hbm_traffic_bench_launch.json
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --butina-n 0 --conformer-mols 0 --cfg1 0"
export NVMK_ROOT=$ROOT
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, hashlib, json, os, sys
out = {}
h = hashlib.sha256()
for name in ("similarity_mfma.hip", "fp4.h", "similarity.hip"):   # same digest as bench.py kernel_source_digest()
    h.update(open(os.path.join(os.environ["NVMK_ROOT"], "nvmolkit_amd", "csrc", name), "rb").read())
out["kernel_source_sha256"] = h.hexdigest()
for line in open(f"{sys.argv[1]}/fetch.log"):          # the bench line of the FETCH pass names the chunk it ran
    if line.startswith("{") and "chunk_rows" in line:
        out["chunk_rows"] = json.loads(line)["config"]["chunk_rows"]
for name, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    vals = []
    for f in glob.glob(f"{sys.argv[1]}/{sub}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "cross_sim_mfma_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    full = sorted(vals)[len(vals) // 8:] if vals else []           # drop the ragged last chunk(s)
    big = [v for v in vals if full and v > 0.9 * max(vals)]
    out[name] = {"launches": len(vals), "full_chunk_launches": len(big), "mean_KiB_per_full_launch": sum(big) / max(len(big), 1)}
out["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 0 --cpu-seconds 0 "
               "--butina-n 0 --conformer-mols 0; full launches = chunk_rows x 1M chunks")
json.dump(out, open(f"{sys.argv[1]}/pmc_hbm_traffic_bench_launch.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
