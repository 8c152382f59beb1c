# per-iteration phase clocks of the MMFF stage, product library against every variant library beside it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for L in "" $(ls nvmolkit_amd/lib/ | sed -n 's/^libnvmolkit_amd_\(.*\)\.so$/\1/p'); do
  LIBP=$GRAFT_REPO_ROOT/nvmolkit_amd/lib/libnvmolkit_amd${L:+_$L}.so
  echo "== ${L:-product}"
  NVMOLKIT_AMD_LIB=$LIBP NVMK_BFGS_PROFILE=1 timeout 300 python tools/bench_conformers.py --mols ${1:-4000} --cache /tmp/nvmk_lib_cache 2>&1 | grep "bfgs profile" | awk '$5 == 2 || ($7+0 > 3000)' | tail -4
done
