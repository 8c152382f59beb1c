#!/usr/bin/env python3
"""Timing of dense-matrix Butina (nvmolkit.clustering.butina; reference benchmarks/butina_clustering_bench.py sizes) on
Tanimoto distance matrices of the synthetic planted-cluster fingerprints.  Usage: python tools/bench_butina_dense.py [N ...]"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from bench import SEED, synth_fingerprints  # noqa: E402
from nvmolkit_amd.clustering import butina  # noqa: E402
from nvmolkit_amd.similarity import crossTanimotoSimilarity  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [10_000, 20_000, 40_000]
dev = torch.device("cuda", 0)
for n in sizes:
    x = synth_fingerprints(n, 64, dev, SEED)
    dist = 1.0 - crossTanimotoSimilarity(x).torch()
    butina(dist[:512, :512].contiguous(), 0.3)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = butina(dist, 0.3)
    labels = res.torch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"n": n, "cutoff": 0.3, "butina_dense_s": dt, "n_clusters": int(labels.max()) + 1}))
