#!/bin/bash
# GPU box: the whole -m gpu suite, smoke(), and the default bench line. usage: bash tools/gpu_suite.sh <tag>
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05_full}
mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests -m gpu -q -s --durations=8 > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|rc |bitwise" $O/pytest.log | tail -12
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 - <<PY
import json
d=json.load(open('$O/bench_default.json'))
c=d['config']
print('f32', round(d['value'],1), d['ms_per_step'], c['kernel_launches_per_step'], 'bf16', c.get('bf16_chips_per_s'), c.get('bf16_kernel_launches_per_step'))
print('ddp1', {k:v for k,v in c.items() if k.startswith('ddp1')})
print('predict', c.get('predict_mpx_per_s'), 'default_point', c.get('default_point_chips_per_s'))
print('roofline', d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('end_to_end_frac'))
PY
