#!/bin/bash
# quick check after a bf16 kernel change: kernel + model parity tests, then the bf16 bench (no extras)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_quick}
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_bf16_kernels_gpu.py tests/test_bf16_model_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
for i in 1 2; do
timeout 300 python3 bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bf16_$i.json 2> $O/bf16_$i.err
python3 -c "
import json; d=json.load(open('$O/bf16_$i.json')); print('bf16', round(d['value'],1), round(d['ms_per_step'],2), d['config']['kernel_launches_per_step'])"
done
