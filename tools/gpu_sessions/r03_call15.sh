#!/bin/bash
# Round 3, call 15: ETKDG batch size x concurrent batches with the final wave threshold (176 coordinates).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call15}
mkdir -p $O
cd $ROOT
timeout 600 python tools/sweep_embed_batch.py --mols 10000 > $O/sweep.jsonl 2> $O/sweep.err; cat $O/sweep.jsonl
