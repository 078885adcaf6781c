"""Queue-occupancy summary of a rocprofv3 --kernel-trace results .db: how much of the wall-clock span the GPU queues
were busy (per queue and their union), how much was idle (launch gaps: what a hipGraph replay removes), and which
kernels sit behind the longest idle gaps.

    python tools/timeline.py <results.db> [tail_fraction]

Only the last `tail_fraction` (default 0.5) of the dispatches is analysed so that warm-up / autotune is excluded.
"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
sub = rows[int(len(rows) * (1.0 - frac)):]
span = max(r[2] for r in sub) - sub[0][1]
byq = collections.defaultdict(float)
for r in sub:
    byq[r[3]] += r[2] - r[1]
ev = sorted((r[1], r[2], r[0]) for r in sub)
union = 0
cs, ce = ev[0][0], ev[0][1]
gaps = collections.defaultdict(lambda: [0, 0.0])
for s, e, name in ev[1:]:
    if s > ce:
        union += ce - cs
        g = gaps[name.split("(")[0][:60]]
        g[0] += 1
        g[1] += s - ce
        cs, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cs
print(f"dispatches {len(sub)}  span {span / 1e6:.2f} ms  union busy {union / 1e6:.2f} ms ({union / span:.1%})  "
      f"idle {(span - union) / 1e6:.2f} ms ({1 - union / span:.1%})")
for q, v in sorted(byq.items()):
    print(f"  queue {q}: busy {v / 1e6:.2f} ms ({v / span:.1%})")
print("largest idle-gap totals by the kernel that FOLLOWS the gap:")
for name, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {t / 1e6:7.2f} ms in {n:5d} gaps (avg {t / n / 1e3:6.1f} us)  {name}")
