cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_morgan_gpu.py tests/test_smiles_ingestion.py tests/test_threads_gpu.py -m gpu -q -x 2>&1 | tail -2
python tools/bench_smiles_ingest.py --repeat 1 2>/dev/null
python tools/bench_smiles_ingest.py --repeat 100 2>/dev/null
python bench.py --steps 1 --warmup 0 --butina-n 0 --conformer-mols 0 --cpu-seconds 2 2>/dev/null | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); g=b['secondary']['cfg1_smiles_to_similarity']; print({k:g[k] for k in ('seconds','smiles_parse_seconds','smiles_to_fingerprints_seconds','fingerprints_per_s','matches_cpu_port')})"
