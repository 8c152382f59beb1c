# XCD-aware tile map in the count kernel: parity of the clustering / similarity suites, Butina timings with and without it
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_clustering_gpu.py tests/test_similarity_gpu.py tests/test_full_size_gpu.py -m gpu -q -x 2>&1 | tail -2
python tools/bench_butina.py 1000000 2>/dev/null
NVMK_COUNT_SUPER=32 python tools/bench_butina.py 1000000 2>/dev/null
python tools/bench_butina.py 1000000 --spread 2>/dev/null
