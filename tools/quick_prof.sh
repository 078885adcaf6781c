#!/bin/bash
# GPU box: rocprofv3 kernel stats of a short bench run (isolated: side stream off), top kernels to stdout + csv.
# usage: bash tools/quick_prof.sh <tag> <f32|bf16> [extra bench args]
set -u
TAG=${1:-q}; P=${2:-bf16}; shift 2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
A=""; [ $P = bf16 ] && A="--dtype bf16"
cd /tmp && export TMPDIR=/tmp
export CN_OVERLAP_WGRAD=0
rocprofv3 --kernel-trace --stats -d $O/stats_iso_$P -o s -- python3 $R/bench.py $A --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > /dev/null 2>&1
python3 $R/tools/prof_db.py $O/stats_iso_$P/s_results.db 400 --csv > $O/${P}_kernel_stats_isolated.csv
unset CN_OVERLAP_WGRAD
rocprofv3 --kernel-trace --stats -d $O/stats_$P -o s -- python3 $R/bench.py $A --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > /dev/null 2>&1
python3 $R/tools/prof_db.py $O/stats_$P/s_results.db 400 --csv > $O/${P}_kernel_stats.csv
python3 $R/tools/timeline.py $O/stats_$P/s_results.db 0.5 > $O/${P}_timeline.txt
rm -rf $O/stats_iso_$P $O/stats_$P
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/${P}_kernel_stats_isolated.csv")))
steps=13
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("isolated kernel ms/step", tot/1e6/steps, "launches/step", sum(int(r['Calls']) for r in rows)/steps)
for r in rows[:32]:
    print(f"{r['Name'][:64]:64s} n/step={int(r['Calls'])/steps:6.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:7.1f}")
PY
