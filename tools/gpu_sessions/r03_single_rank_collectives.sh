#!/bin/bash
# The multi-rank code of bench.py on a one-GPU box: one rank under torch.distributed.run with the collectives forced on
# (RCCL init, all-gather of the reference block, barriers, max / sum all-reduces, library generation before the HIP context).
set -u
out=gpurun_out/${1:-r03_single_rank}
mkdir -p "$out"
export NVMK_BENCH_SINGLE_RANK_COLLECTIVES=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 1 --steps 2 --warmup 1 --cpu-seconds 0 --butina-n 0 --cfg1 0 --conformer-mols 1000 \
  > "$out/bench_single_rank_collectives.json" 2> "$out/bench_single_rank_collectives.err"
echo "rc=$?"
tail -c 1500 "$out/bench_single_rank_collectives.json"
tail -5 "$out/bench_single_rank_collectives.err"
