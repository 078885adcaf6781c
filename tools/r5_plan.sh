#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
for v in 1 0; do
  echo "CN_NATIVE_PLAN=$v"
  CN_NATIVE_PLAN=$v timeout 300 python3 tools/small_batch.py bf16 4 2>&1 | grep -v amdgpu | tail -4
  CN_NATIVE_PLAN=$v timeout 300 python3 tools/small_batch.py bf16 4 64 0.1 2>&1 | grep -v amdgpu | tail -4
done
timeout 600 python3 -m pytest tests/test_replay_train_gpu.py -q 2>&1 | tail -2
