#!/bin/bash
# Large systems (150-400 atoms) with the library before the inverse-Hessian pass rework and after it, on one box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_large_ab}
mkdir -p $O
cd $ROOT
for i in 1 2; do
  echo prev >> $O/ab.txt
  NVMOLKIT_AMD_LIB=$ROOT/nvmolkit_amd/lib/libnvmolkit_amd_prev.so timeout 300 python tools/bench_mixed_sizes.py --mols 2000 2>/dev/null | cut -c1-260 >> $O/ab.txt
  echo new >> $O/ab.txt
  timeout 300 python tools/bench_mixed_sizes.py --mols 2000 2>/dev/null | cut -c1-260 >> $O/ab.txt
done
cat $O/ab.txt
