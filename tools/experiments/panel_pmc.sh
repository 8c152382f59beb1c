# PMC counters of the symmetric all-pairs pass at 1M, tile against panel kernel: separate --pmc passes, no tracing
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_panel
mkdir -p $O
for T in ${KERNELS:-panel tile}; do
  for P in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    tag=$(echo $P | cut -d' ' -f1)
    rm -rf /tmp/pmc_${T}_$tag
    NVMK_COUNT_KERNEL=$T timeout 300 rocprofv3 --pmc $P -f csv -d /tmp/pmc_${T}_$tag -- python $R/tools/bench_butina.py 1000000 > $O/pmc_${T}_$tag.log 2>&1
  done
done
python - "$O" <<'PY'
import csv, glob, json, sys
out = {}
for d in glob.glob("/tmp/pmc_*"):
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "neighbor_count" not in k: continue
            name = ("panel" if "panel_kernel" in k else "tile_sym" if "<0, true, true>" in k else "tile_rect")
            out.setdefault(name, {}).setdefault(r["Counter_Name"], 0.0)
            out[name][r["Counter_Name"]] += float(r["Counter_Value"])
json.dump(out, open(sys.argv[1] + "/pmc_counts.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
