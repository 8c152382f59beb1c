cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_e2e
# (the library of the whole file: generated once by the benchmark itself, ~30 s)
python - <<'PY'
import pickle, sys
sys.path.insert(0, '.')
from pathlib import Path
from nvmolkit_amd import synthetic
lib, _ = synthetic.smiles_file_library(Path('tests/golden/chembl_10k.smi'), n_mols=10000, max_atoms=100000)
pickle.dump(lib, open('/tmp/chembl_all.pkl', 'wb'), protocol=pickle.HIGHEST_PROTOCOL)
PY
timeout 900 python tools/experiments/e2e_overlap_probe.py /tmp/chembl_all.pkl 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_e2e/probe.txt
