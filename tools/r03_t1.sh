#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_bf16_model_gpu.py -m gpu -q 2>&1 | tail -4
for P in bf16 f32; do
A=""; [ $P = bf16 ] && A="--dtype bf16"
for i in 1 2; do
timeout 300 python3 bench.py $A --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('$P', round(d['value'],1), round(d['ms_per_step'],2))"
done
done
