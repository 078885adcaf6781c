"""End-of-step tail of a rocprofv3 --kernel-trace results .db: for every optimizer launch (cn_adamw_kernel), the idle
time of the compute queue in front of it, i.e. how long the step waited for the weight-gradient side stream after the
last kernel of the data-gradient chain.

    python tools/tail.py <results.db>
"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
byq = collections.Counter(r[3] for r in rows)
main_q = byq.most_common(1)[0][0]
adam = [r for r in rows if "cn_adamw" in r[0]]
print(f"queues: {dict(byq)}  compute queue = {main_q}  optimizer launches = {len(adam)}")
prev_adam_end = None
for a in adam[-8:]:
    before = [r for r in rows if r[2] <= a[1] and (prev_adam_end is None or r[1] >= prev_adam_end)]
    main = [r for r in before if r[3] == main_q and "cn_sumsq" not in r[0] and "cn_adamw" not in r[0]]
    side = [r for r in before if r[3] != main_q]
    if main and side:
        # last kernel of the backward chain on the compute queue that is NOT part of the clip / optimizer epilogue
        last_main = max(main, key=lambda r: r[2])
        last_side = max(side, key=lambda r: r[2])
        step = a[2] - (prev_adam_end or before[0][1])
        print(f"step {step / 1e6:7.2f} ms  last compute-queue kernel ends {(a[1] - last_main[2]) / 1e3:7.1f} us before adamw "
              f"({last_main[0][:40]}), last side-queue kernel {(a[1] - last_side[2]) / 1e3:7.1f} us before "
              f"({last_side[0][:40]}); side work after the last compute kernel: "
              f"{sum(min(r[2], a[1]) - max(r[1], last_main[2]) for r in side if r[2] > last_main[2]) / 1e3:7.1f} us")
    prev_adam_end = a[2]
