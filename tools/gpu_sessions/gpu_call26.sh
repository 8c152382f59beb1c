# XCD-aware tile map of the dense kernel: parity, bench twice, pure-store rate of the same box, FETCH/WRITE of the launches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_call26; mkdir -p $O
timeout 900 python -m pytest tests/test_similarity_gpu.py tests/test_full_size_gpu.py -m gpu -q -x 2>&1 | tail -2
tools/ubench_store 2>&1 | head -2
for i in 1 2; do python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --butina-n 0 --conformer-mols 0 --cfg1 0 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['value'], b['roofline']['frac'], b['roofline']['avg_launch_ms'])"; done
bash tools/profile_bench_traffic.sh 2>&1 | grep -A3 "FETCH_SIZE\|WRITE_SIZE" | grep mean
