#!/bin/bash
# after a bf16 kernel change: parity tests of the bf16 kernels + model, then the bf16 bench twice
set -u
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_bf16_kernels_gpu.py tests/test_bf16_model_gpu.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2 3; do
timeout 300 python3 bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('bf16', round(d['value'],1), round(d['ms_per_step'],2))"
done
