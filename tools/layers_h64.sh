#!/bin/bash
# per-launch contraction table + kernel stats of the bf16 step at hidden 64 (batch given, default 4: the reference CLI's point)
set -u
B=${1:-4}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${2:-h64}
mkdir -p $O
cd $R
rm -f /tmp/d.tsv
CN_OVERLAP_WGRAD=0 CN_PROF_DUMP=/tmp/d.tsv timeout 300 python3 bench.py --dtype bf16 --hidden 64 --batch $B --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/b.json 2> $O/b.err
python3 tools/layerprof.py /tmp/d.tsv 6 bf16 > $O/layers_bf16_h64_b$B.txt
head -60 $O/layers_bf16_h64_b$B.txt
cd /tmp && export TMPDIR=/tmp
CN_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats -d $O/st -o s -- python3 $R/bench.py --dtype bf16 --hidden 64 --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 $R/tools/prof_db.py $O/st/s_results.db 400 --csv > $O/kernel_stats_iso_b$B.csv
rm -rf $O/st
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_iso_b$B.csv")))
steps=13
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("isolated kernel ms/step", tot/1e6/steps, "launches/step", sum(int(r['Calls']) for r in rows)/steps)
for r in rows[:30]:
    print(f"{r['Name'][:64]:64s} n/step={int(r['Calls'])/steps:6.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:7.1f}")
PY
