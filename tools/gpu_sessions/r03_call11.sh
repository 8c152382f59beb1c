#!/bin/bash
# Round 3, call 11: rocprofv3 kernel trace + PMC passes of the conformer pipeline with the hybrid BFGS (where does ETKDG's wall time go?)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call11}
mkdir -p $O
cd $ROOT
bash tools/profile_conformers.sh $(basename $O)/prof 2000 auto > $O/profile.log 2>&1
tail -5 $O/profile.log
python - $O/prof <<'P'
import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(f"{d}/trace_auto/**/*_kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows)
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in ev:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, n in ev:
        per[n][0] += e - s; per[n][1] += 1
    print("span_ms", (t1 - t0) / 1e6, "gpu_busy_union_ms", busy / 1e6, "kernels", len(ev))
    for n, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"{t/1e6:10.2f} ms {c:6d} calls  {n[:110]}")
    # gaps > 0.5 ms between union-busy intervals
P
