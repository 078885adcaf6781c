#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_hiptrace}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats -d $O/t -o h -- python3 $R/bench.py --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/out.txt 2>&1
ls $O/t | head
python3 - <<PY
import sqlite3,glob
db=glob.glob("$O/t/*results.db")[0]
c=sqlite3.connect(db)
print("copies by (name, size):")
for row in c.execute("select name, size, count(*), stream_name from memory_copies group by name, size, stream_name order by count(*) desc limit 25"): print(row)
print("top:")
try:
    cols=[r[1] for r in c.execute("pragma table_info(top)")]
    print(cols)
    for row in c.execute("select * from top limit 40"): print(row)
except Exception as e: print(e)
PY
rm -rf $O/t
