#!/bin/bash
# Wave priority by size class: mixed-size batches and the 10 000-molecule bench with the library before and after, on one box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_prio_ab}
mkdir -p $O
cd $ROOT
( timeout 300 python -m pytest tests/test_bfgs_parity_gpu.py -m gpu -q -x -k "mixed or large or trajectory" 2>&1 | tail -2 ) | tee $O/pytest.txt
for i in 1 2; do
  echo prev >> $O/ab.txt
  NVMOLKIT_AMD_LIB=$ROOT/nvmolkit_amd/lib/libnvmolkit_amd_prev.so timeout 300 python tools/bench_mixed_sizes.py --mols 2000 2>/dev/null | cut -c1-260 | head -3 >> $O/ab.txt
  NVMOLKIT_AMD_LIB=$ROOT/nvmolkit_amd/lib/libnvmolkit_amd_prev.so timeout 300 python tools/bench_conformers.py --mols 10000 2>/dev/null | tail -1 | cut -c1-420 >> $O/ab.txt
  echo new >> $O/ab.txt
  timeout 300 python tools/bench_mixed_sizes.py --mols 2000 2>/dev/null | cut -c1-260 | head -3 >> $O/ab.txt
  timeout 300 python tools/bench_conformers.py --mols 10000 2>/dev/null | tail -1 | cut -c1-420 >> $O/ab.txt
done
cat $O/ab.txt
