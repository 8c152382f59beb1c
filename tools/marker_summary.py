#!/usr/bin/env python3
"""Summarise a `rocprofv3 --marker-trace` CSV of the library's roctx ranges: per range name the number of calls and the total /
mean wall clock, then the stage timeline of the first full ETKDG batch (start and length of every stage inside its
"ETKDG batch" range).  Usage: python tools/marker_summary.py marker_api_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
name_key = next(k for k in rows[0] if k.lower() in ("function", "name", "message"))
start_key = next(k for k in rows[0] if "start" in k.lower())
end_key = next(k for k in rows[0] if "end" in k.lower())
agg = collections.defaultdict(lambda: [0, 0.0])
spans = []
for r in rows:
    t0, t1 = int(r[start_key]), int(r[end_key])
    agg[r[name_key]][0] += 1
    agg[r[name_key]][1] += (t1 - t0) * 1e-6
    spans.append((t0, t1, r[name_key]))
print(f"{'range':<70}{'calls':>8}{'total ms':>12}{'mean ms':>12}")
for name, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{name[:69]:<70}{n:>8d}{ms:>12.3f}{ms / n:>12.3f}")
batches = sorted(s for s in spans if s[2] == "ETKDG batch")
if batches:
    b0, b1, _ = max(batches, key=lambda s: s[1] - s[0])
    print(f"\nlongest ETKDG batch: {(b1 - b0) * 1e-6:.3f} ms; ranges inside it (start ms, length ms):")
    for t0, t1, name in sorted(spans):
        if b0 <= t0 and t1 <= b1 and name != "ETKDG batch":
            print(f"  {(t0 - b0) * 1e-6:>10.3f} {(t1 - t0) * 1e-6:>10.3f}  {name}")
