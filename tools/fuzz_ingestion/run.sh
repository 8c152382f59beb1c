#!/bin/bash
# builds the fuzzer with AddressSanitizer + UBSan against the library's host sources and runs it over the golden inputs
set -eu
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$HERE/../..
OUT=${TMPDIR:-/tmp}/nvmk_fuzz_ingestion
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -D__HIP_PLATFORM_AMD__ \
  -I/opt/rocm/include -I$ROOT/include $HERE/fuzz_ingestion.cpp $ROOT/nvmolkit_amd/csrc/smiles.cpp $ROOT/nvmolkit_amd/csrc/runtime.cpp \
  -o $OUT -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread
$OUT smi $ROOT/tests/golden/chembl_1k.smi ${1:-200}
$OUT smi $ROOT/tests/golden/more_rdkit_smiles.smi ${1:-200}
$OUT sdf $ROOT/tests/golden/MMFF94_dative_every4th.sdf ${1:-200}
$OUT sdf $ROOT/tests/golden/larger_molecules.sdf ${1:-200}
