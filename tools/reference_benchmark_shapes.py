#!/usr/bin/env python3
"""The shapes of the reference's own Python benchmarks, run on this library without RDKit (SMILES are ingested by the
library itself from the reference's benchmark file, tests/golden/chembl_10k.smi):

* cross similarity  — benchmarks/cross_similarity_bench.py:26-84: N x N Tanimoto (and cosine with --cosine) of Morgan
  radius-3 / 1024-bit fingerprints for N = 2000 ... 32000 (the molecule list repeated to the largest size), GPU time only;
* Butina            — benchmarks/butina_clustering_bench.py: Morgan radius-2 / 1024-bit, cutoffs 1e-10 / 0.1 / 0.2 / 0.35 /
  1.1 (edge cutoffs only up to 20 000 rows), dense `butina` on a precomputed distance matrix and matrix-free `fused_butina`
  (cutoffs above 1 are not accepted by fused_butina, in the reference either).

One JSON line per measurement (mean / std in ms over --runs timed repetitions after one warm-up).
    python tools/reference_benchmark_shapes.py [--what similarity,butina] [--runs 3] [--max-size 32000] [--cosine]
    python tools/reference_benchmark_shapes.py --dry-run        # host side only: ingestion and the plan, no GPU needed
"""
import argparse
import json
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from nvmolkit_amd.fingerprints import MorganFingerprintGenerator, SmilesSet  # noqa: E402

SIM_SIZES = [2000, 4000, 6000, 8000, 10000, 12000, 14000, 16000, 20000, 24000, 28000, 32000]
BUTINA_SIZES = [1000, 5000, 10000, 20000, 30000, 40000]
CUTOFFS = [1e-10, 0.1, 0.2, 0.35, 1.1]


def timed(fn, runs: int):
    fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(runs):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t) * 1e3)
    return statistics.mean(ms), (statistics.stdev(ms) if len(ms) > 1 else 0.0)


def repeat_rows(fps: torch.Tensor, n: int) -> torch.Tensor:
    reps = (n + fps.shape[0] - 1) // fps.shape[0]
    return fps.repeat(reps, 1)[:n].contiguous()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", default=str(ROOT / "tests" / "golden" / "chembl_10k.smi"))
    ap.add_argument("--what", default="similarity,butina")
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--max-size", type=int, default=40000)
    ap.add_argument("--cosine", action="store_true")
    ap.add_argument("--dry-run", action="store_true")
    args = ap.parse_args()
    what = set(args.what.split(","))
    t = time.perf_counter()
    mols = SmilesSet.from_file(args.input)
    keep = np.flatnonzero((mols.status == 0) & (mols.n_atoms > 0))
    print(json.dumps({"input": args.input, "lines": len(mols), "ingested": int(len(keep)), "parse_ms": (time.perf_counter() - t) * 1e3}))
    sim_sizes = [n for n in SIM_SIZES if n <= args.max_size]
    butina_sizes = [n for n in BUTINA_SIZES if n <= args.max_size]
    if args.dry_run:
        print(json.dumps({"plan": {"similarity": sim_sizes if "similarity" in what else [], "butina": butina_sizes if "butina" in what else [],
                                   "cutoffs": CUTOFFS}}))
        return
    from nvmolkit_amd.clustering import butina, fused_butina
    from nvmolkit_amd.similarity import crossCosineSimilarity, crossTanimotoSimilarity

    smiles = [line.split()[0] for line in Path(args.input).read_text().splitlines() if line.strip()]
    smiles = [smiles[i] for i in keep]
    if "similarity" in what:
        fps = MorganFingerprintGenerator(radius=3, fpSize=1024).GetFingerprintsFromSmiles(smiles).torch()
        for name, fn in [("tanimoto", crossTanimotoSimilarity)] + ([("cosine", crossCosineSimilarity)] if args.cosine else []):
            for n in sim_sizes:
                x = repeat_rows(fps, n)
                mean, std = timed(lambda: fn(x), args.runs)
                print(json.dumps({"bench": f"nvmolkit_gpu-only_{name}sim_fpsize_1024_{n}mols", "mean_ms": mean, "std_ms": std,
                                  "pairs_per_s": n * n / (mean * 1e-3)}), flush=True)
                del x
    if "butina" in what:
        fps = MorganFingerprintGenerator(radius=2, fpSize=1024).GetFingerprintsFromSmiles(smiles).torch()
        torch.manual_seed(20260926)
        for n in butina_sizes:
            # beyond the molecules of the file the reference pads with random fingerprints / random distances
            # (butina_clustering_bench.py:71-96); the same here
            real = min(n, fps.shape[0])
            x = torch.randint(-(2**31), 2**31 - 1, (n, fps.shape[1]), dtype=torch.int32, device=fps.device)
            x[:real] = fps[:real]
            dist = torch.rand(n, n, dtype=torch.float64, device=fps.device)
            dist = torch.abs(dist - dist.T).clip(0.01, 0.99)
            dist.fill_diagonal_(0.0)
            dist[:real, :real] = 1.0 - crossTanimotoSimilarity(fps[:real].contiguous()).torch()
            for cutoff in CUTOFFS:
                if cutoff in (1e-10, 1.1) and n > 20000:
                    continue
                row = {"bench": "butina", "size": n, "cutoff": cutoff}
                mean, std = timed(lambda: butina(dist, cutoff).torch(), args.runs)
                row.update(nvmolkit_time_ms=mean, nvmolkit_std_ms=std)
                if cutoff <= 1.0:
                    mean, std = timed(lambda: fused_butina(x, cutoff=cutoff, metric="tanimoto"), args.runs)
                    row.update(fused_butina_time_ms=mean, fused_butina_std_ms=std)
                print(json.dumps(row), flush=True)
            del dist, x
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
