#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_v16}
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_bf16_kernels_gpu.py -m gpu -q -x > $O/pytest_k.log 2>&1
echo "rc $?" >> $O/pytest_k.log
grep -E "^FAILED|passed|failed|^E  " $O/pytest_k.log | head -12
timeout 900 python3 -m pytest tests/test_bf16_model_gpu.py tests/test_predict_edges_gpu.py tests/test_model_gpu.py -m gpu -q > $O/pytest_m.log 2>&1
echo "rc $?" >> $O/pytest_m.log
grep -E "^FAILED|passed|failed" $O/pytest_m.log | head -12
for V in 1 0 1 0; do
CN_BCONV_V16=$V timeout 300 python3 bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bf16_v$V.json 2> $O/bf16_v$V.err
python3 -c "
import json; d=json.load(open('$O/bf16_v$V.json')); print('v16=$V bf16', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['kernel'], round(d['roofline']['achieved'],1))"
done
