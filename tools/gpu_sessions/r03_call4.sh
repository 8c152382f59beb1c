#!/bin/bash
# Round 3, call 4: parallel Butina rounds — parity (all variants), full size, timing serial vs parallel.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call4}
mkdir -p $O
cd $ROOT
( time timeout 900 python -m pytest tests/test_clustering_gpu.py tests/test_benchmark_molecules_gpu.py tests/test_full_size_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
tail -15 $O/pytest.log
python - > $O/butina_ab.jsonl 2> $O/butina_ab.err <<'P'
import json, sys, time
sys.path.insert(0, '.')
import torch
from bench import SEED, synth_fingerprints
from nvmolkit_amd import _native
from nvmolkit_amd.clustering import fused_butina
dev = torch.device("cuda", 0)
for n in (100_000, 1_000_000):
    x = synth_fingerprints(n, 64, dev, SEED)
    for mode in (None, "serial"):
        with _native.options(NVMK_BUTINA_ROUNDS=mode):
            best = None
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                clusters, sizes = fused_butina(x, 0.3)
                t = time.perf_counter() - t0
                best = t if best is None else min(best, t)
            print(json.dumps({"n": n, "rounds": mode or "parallel", "fused_butina_s": best, "n_clusters": len(clusters)}), flush=True)
            if mode is None: ref = clusters
            else: print(json.dumps({"n": n, "identical_to_serial": ref == clusters}), flush=True)
P
cat $O/butina_ab.jsonl; tail -3 $O/butina_ab.err
