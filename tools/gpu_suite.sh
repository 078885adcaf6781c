#!/bin/bash
# GPU box: the whole -m gpu suite, smoke(), and two short bf16 / fp32 bench lines. usage: bash tools/gpu_suite.sh <tag>
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04_full}
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|rc " $O/pytest.log | tail -12
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2; do
timeout 300 python3 bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bf16_$i.json 2> $O/bf16_$i.err
python3 -c "
import json; d=json.load(open('$O/bf16_$i.json')); print('bf16', round(d['value'],1), round(d['ms_per_step'],2), d['config']['kernel_launches_per_step'])"
done
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/f32.json 2> $O/f32.err
python3 -c "
import json; d=json.load(open('$O/f32.json')); print('f32', round(d['value'],1), round(d['ms_per_step'],2), d['config']['kernel_launches_per_step'])"
