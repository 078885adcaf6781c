"""One training step of a rocprofv3 --kernel-trace results .db as an ordered list: start (us from the step's first
kernel), duration, queue, kernel name. The step = the dispatches between the last two cn_adamw_kernel launches.

    python tools/step_trace.py <results.db> > step.txt
"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
ad = [i for i, r in enumerate(rows) if r[0].startswith("cn_adamw_kernel")]
a, b = ad[-2], ad[-1]
t0 = rows[a + 1][1]
qs = sorted({r[3] for r in rows[a + 1:b + 1]})
for r in rows[a + 1:b + 1]:
    print(f"{(r[1] - t0) / 1e3:10.1f} {(r[2] - r[1]) / 1e3:8.1f} q{qs.index(r[3])} {r[0].split('(')[0][:90]}")
print(f"# step span {(rows[b][2] - t0) / 1e3:.1f} us, {b - a} dispatches")
