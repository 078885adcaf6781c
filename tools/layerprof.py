"""Aggregate a CN_PROF_DUMP file (one line per contraction launch) by launch description.

usage: CN_PROF_DUMP=/tmp/d.tsv python bench.py --steps 3 --no-cpu-baseline; python tools/layerprof.py /tmp/d.tsv 3 [f32|bf16]

The "lost" column is the launch time beyond its time at the MFMA peak of the launch's OWN precision (f32-input MFMA 157.3
TFLOP/s, dense bf16 2500): bf16 launches (descriptions starting "bconv" / "bwgrad") are priced against the bf16 peak.
"""
import sys
from collections import defaultdict

path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = defaultdict(lambda: [0.0, 0.0, 0])
for line in open(path):
    kind, desc, us, fl = line.rstrip("\n").split("\t")
    a = agg[desc]
    a[0] += float(us); a[1] += float(fl); a[2] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in agg.values())
print(f"total {tot / steps / 1e3:.2f} ms/step over {len(rows)} distinct launches")
for desc, (us, fl, n) in rows:
    tf = fl / us / 1e6 if us > 0 else 0.0
    bf16 = desc.startswith(("bconv", "bwgrad")) or (len(sys.argv) > 3 and sys.argv[3] == "bf16")
    ideal = fl / (2500e6 if bf16 else 157.3e6)  # us at the MFMA peak of the launch's precision
    print(f"{us / steps:9.1f} us/step  n={n / steps:5.1f}  {us / n:8.1f} us  {tf:6.1f} TF  lost {(us - ideal) / steps:8.1f} us/step  {desc}")
