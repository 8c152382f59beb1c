# kernel times of the fused-Butina all-pairs pass at 1M: tile kernel against the row-panel kernel (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_panel
mkdir -p $O
for T in tile panel; do
  rm -rf /tmp/prof_$T
  NVMK_COUNT_KERNEL=$T timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$T -- python $R/tools/bench_butina.py 1000000 > $O/probe_$T.log 2>&1
  f=$(find /tmp/prof_$T -name '*kernel_stats.csv' | head -1)
  echo "== $T"; head -8 $f | cut -c1-200
  cp $f $O/kernel_stats_$T.csv
done
