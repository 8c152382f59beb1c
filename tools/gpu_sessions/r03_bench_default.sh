#!/bin/bash
# The default bench.py line (what the driver runs at round end), its stderr and its wall time.
set -u
out=gpurun_out/${1:-r03_bench_default}
mkdir -p "$out"
t0=$(date +%s)
timeout 900 python bench.py > "$out/bench.json" 2> "$out/bench.err"
echo "rc=$? seconds=$(( $(date +%s) - t0 ))"
tail -c 600 "$out/bench.json"; echo
grep -v amdgpu.ids "$out/bench.err" | tail -5
