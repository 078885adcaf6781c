#!/bin/bash
# GPU box: kernel stats + timeline of the sliding-window predict (36 windows per batch), and the unprofiled timing.
# usage: bash tools/predict_profile.sh <tag> [bf16|f32]
set -u
TAG=${1:-pp}; P=${2:-bf16}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
python3 $R/tools/predict_prof.py $P 20 36 > $O/predict_${P}_timing.txt 2>&1
python3 $R/tools/predict_prof.py $P 1 36 >> $O/predict_${P}_timing.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/st -o p -- python3 $R/tools/predict_prof.py $P 10 36 > /dev/null 2>&1
python3 $R/tools/prof_db.py $O/st/p_results.db 400 --csv > $O/predict_${P}_kernel_stats.csv
python3 $R/tools/timeline.py $O/st/p_results.db 0.5 > $O/predict_${P}_timeline.txt
rm -rf $O/st
cat $O/predict_${P}_timing.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/predict_${P}_kernel_stats.csv")))
n=12
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("kernel ms/scene", tot/1e6/n, "launches/scene", sum(int(r['Calls']) for r in rows)/n)
for r in rows[:40]:
    print(f"{r['Name'][:70]:70s} n={int(r['Calls'])/n:6.1f} ms={float(r['TotalDurationNs'])/1e6/n:7.3f} avg_us={float(r['AverageNs'])/1e3:7.1f}")
PY
head -20 $O/predict_${P}_timeline.txt
