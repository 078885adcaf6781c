#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
for M in normal low; do
  CN_SIDE_STREAM=$M timeout 200 python3 -m pytest tests/test_ddp_engine_gpu.py -m gpu -q -x > $O/ddp_$M.log 2>&1
  echo "rc $?" >> $O/ddp_$M.log
  tail -3 $O/ddp_$M.log
done
timeout 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_bf16_kernels_gpu.py tests/test_model_gpu.py tests/test_bf16_model_gpu.py -m gpu -q --durations=25 > $O/dur.log 2>&1
grep -A 30 "slowest" $O/dur.log | head -40
tail -3 $O/dur.log
