cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_hh; mkdir -p $O
NVMOLKIT_AMD_LIB=$GRAFT_REPO_ROOT/nvmolkit_amd/lib/libnvmolkit_amd_hh.so timeout 600 python -m pytest tests/test_bfgs_parity_gpu.py -m gpu -q -k "team" -p no:cacheprovider > $O/team_tests_hh.log 2>&1; tail -2 $O/team_tests_hh.log
bash tools/gpu_session.sh r06_hh ab_chembl_all
