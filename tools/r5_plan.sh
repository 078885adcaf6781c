#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_replay_train_gpu.py tests/test_predict_edges_gpu.py -q -x 2>&1 | tail -6
for v in 1 0; do
  echo "CN_NATIVE_PLAN=$v"
  CN_NATIVE_PLAN=$v timeout 300 python3 bench.py --child default_point --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms','kernel_launches_per_step','speedup_vs_eager')}, 'eager', d['eager']['value'], d['eager']['host_enqueue_ms'])"
  CN_NATIVE_PLAN=$v timeout 300 python3 tools/small_batch.py bf16 4 2>&1 | tail -3
  CN_NATIVE_PLAN=$v timeout 300 python3 tools/predict_prof.py bf16 20 4 2>&1 | tail -1
  CN_NATIVE_PLAN=$v timeout 300 python3 tools/predict_prof.py bf16 20 36 2>&1 | tail -1
done
