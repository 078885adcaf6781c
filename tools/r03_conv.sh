#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd $R
CN_LIB_PATH=cultionet_amd/csrc/libcultionet_hip_stamp.so python3 tools/bconv_stamps.py 32 128 100 100 128 3 2>&1 | grep -A4 "slot 0"
python3 tools/bconv_bench.py 2>&1 | tail -12
timeout 600 python3 -m pytest tests/test_bf16_kernels_gpu.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do
timeout 300 python3 bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('bf16', round(d['value'],1), round(d['ms_per_step'],2))"
done
