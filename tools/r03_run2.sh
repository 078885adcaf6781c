#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc $?" >> $O/bench_default.err
tail -8 $O/pytest.log
python3 - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print("default", d["value"], d["ms_per_step"], "bf16", d.get("bf16",{}).get("value"), "predict", d.get("predict",{}).get("value"), "feed", d.get("feed",{}).get("value"))
print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["launches_per_step"])
PY
