# Elimination experiment on the row-panel count kernel through the stand-alone pass (nvmk_neighbor_counts of a set against itself:
# tools/bench_butina.py --skip-butina), product library against variant libraries built with -DNVMK_PANEL_SKIP=<bits>
# (1 no epilogue, 2 no DMA and no wait for it in the loop, 4 no barrier, 8 no fragment reads; the knobs are not in the tree: see
# tools/experiments/README.md).  The variants' counts are meaningless; only the times are read.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_panel
mkdir -p $O
: > $O/elimination.txt
for rep in 1 2; do
for L in "" $(ls $R/nvmolkit_amd/lib/ | sed -n 's/^libnvmolkit_amd_\(.*\)\.so$/\1/p'); do
  LIBP=$R/nvmolkit_amd/lib/libnvmolkit_amd${L:+_$L}.so
  echo -n "${L:-product} " | tee -a $O/elimination.txt
  NVMOLKIT_AMD_LIB=$LIBP NVMK_COUNT_KERNEL=panel timeout -k 5 60 python $R/tools/bench_butina.py 1000000 --skip-butina 2>/dev/null | tail -1 | cut -c1-120 | tee -a $O/elimination.txt
done
done
