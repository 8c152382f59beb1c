cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_device_chain_gpu.py tests/test_forcefield_gpu.py tests/test_config_size_gpu.py -m gpu -q -x 2>&1 | tail -2
python tools/bench_conformers.py --mols 2000 2>/dev/null
python tools/bench_conformers.py --mols 10000 2>/dev/null
