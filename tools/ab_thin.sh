#!/bin/bash
# same-box A/B of two builds of the direct head convolutions: LIBS="libcultionet_hip.so libcultionet_hip_b2.so"
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests/test_kernels_gpu.py -q -k thin 2>&1 | tail -1
for round in 1 2; do
  for LIB in ${LIBS:-libcultionet_hip.so}; do
    export CN_LIB_PATH=$R/cultionet_amd/csrc/$LIB
    echo "$LIB: $(python3 tools/thin_bench.py 8 128 100 2>&1 | grep -v amdgpu | awk '{printf "%s %s %s | ", $1, $2, $3}')"
  done
done
for round in 1 2; do
  for LIB in ${LIBS:-libcultionet_hip.so}; do
    export CN_LIB_PATH=$R/cultionet_amd/csrc/$LIB
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$LIB f32', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
