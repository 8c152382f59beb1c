cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06_twice}; mkdir -p $O
for i in 1 2; do
  timeout 900 python -m pytest tests/test_bfgs_parity_gpu.py tests/test_hessian_update_gpu.py -m gpu -q -p no:cacheprovider > $O/run$i.log 2>&1
  tail -3 $O/run$i.log | cut -c1-200
done
