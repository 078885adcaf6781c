#!/bin/bash
# round 3, first GPU pass: side-stream modes, GPU test suite, the default bench line
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
for M in normal low mask:192 mask:128; do
  for P in f32 bf16; do
    A=""; [ $P = bf16 ] && A="--dtype bf16"
    CN_SIDE_STREAM=$M timeout 300 python3 bench.py $A --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/side_${M/:/_}_$P.json 2> $O/side_${M/:/_}_$P.err
  done
done
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc $?" >> $O/bench_default.err
tail -3 $O/pytest.log
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/side_*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), round(d["value"],1), round(d["ms_per_step"],2))
    except Exception as e: print(f, "ERR", e)
try:
    d=json.load(open("$O/bench_default.json"))
    print("default", d["value"], d["ms_per_step"], "bf16", d.get("bf16",{}).get("value"), "predict", d.get("predict",{}).get("value"), "feed", d.get("feed"))
except Exception as e: print("default ERR", e)
PY
