"""Print the per-kernel summary of a rocprofv3 results .db (rocpd sqlite): name, calls, total ms, avg us, %."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot / 1e3:.1f} ms over {sum(r[1] for r in rows)} launches", file=sys.stderr)
if "--csv" in sys.argv:
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage"')
    for r in rows:
        print(f'"{r[0]}",{r[1]},{int(r[2] * 1e3)},{r[3] * 1e3:.1f},{r[4]:.2f}')
else:
    for r in rows[:n]:
        print(f"{r[2] / 1e3:9.2f} ms {r[1]:6d} x {r[3]:8.1f} us {r[4]:5.1f}%  {r[0][:100]}")
