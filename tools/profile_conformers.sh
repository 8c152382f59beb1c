#!/bin/bash
# Runs on the GPU box (via gpurun): kernel statistics and PMC counters of the conformer pipeline (ETKDG + MMFF on the
# synthetic drug-like set).  PMC passes are separate rocprofv3 runs and never combined with tracing.
# Usage: tools/profile_conformers.sh <out-dir-under-gpurun_out> [mols] [NVMK_BFGS_LDS policy ...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-prof_conformers}
MOLS=${2:-1000}
shift 2 || true
POLICIES=${@:-auto}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/tools/bench_conformers.py --mols $MOLS"
for pol in $POLICIES; do
  export NVMK_BFGS_LDS=$pol
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_$pol -- $BENCH > $OUT/trace_$pol.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch_$pol -- $BENCH > $OUT/pmc_fetch_$pol.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write_$pol -- $BENCH > $OUT/pmc_write_$pol.log 2>&1
done
unset NVMK_BFGS_LDS
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -f csv \
  -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
python - "$OUT" $POLICIES <<'PY'
import collections, csv, glob, json, sys
out_dir, policies = sys.argv[1], sys.argv[2:]
summary = {}
def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"{out_dir}/{sub}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            agg[name]["dispatches_" + r["Counter_Name"]] += 1
    return agg
for pol in policies:
    s = {}
    for f in glob.glob(f"{out_dir}/trace_{pol}/**/*_kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "bfgs_kernel" in r["Name"] or "energy_kernel" in r["Name"]:
                s.setdefault(r["Name"].split("(")[0], {})["stats"] = {k: r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage")}
    for sub, cname in ((f"pmc_fetch_{pol}", "FETCH_SIZE"), (f"pmc_write_{pol}", "WRITE_SIZE")):
        for name, c in counters(sub).items():
            if "bfgs_kernel" in name:
                s.setdefault(name, {})[cname + "_KiB_total"] = c[cname]
    for name, v in s.items():
        if "FETCH_SIZE_KiB_total" in v and "WRITE_SIZE_KiB_total" in v and "stats" in v:
            hbm = (2.0 * v["FETCH_SIZE_KiB_total"] + v["WRITE_SIZE_KiB_total"]) * 1024.0
            v["hbm_bytes_2xFETCH_plus_WRITE"] = hbm
            v["hbm_GBps_over_kernel_time"] = hbm / float(v["stats"]["TotalDurationNs"])
    summary[pol] = s
sq = {}
for name, c in counters("pmc_sq").items():
    if "bfgs_kernel" in name:
        sq[name] = {k: v for k, v in c.items() if not k.startswith("dispatches_")}
        w = c.get("SQ_WAVE_CYCLES", 0.0)
        if w:
            for k in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS"):
                sq[name][k + "/SQ_WAVE_CYCLES"] = c.get(k, 0.0) / w
summary["sq_counters_default_policy"] = sq
summary["note"] = ("rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (separate passes) -- python "
                   "tools/bench_conformers.py; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)")
json.dump(summary, open(f"{out_dir}/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1)[:6000])
PY
