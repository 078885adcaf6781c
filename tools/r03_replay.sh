#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python3 -m pytest tests/test_predict_edges_gpu.py -m gpu -q 2>&1 | tail -6
for bs in 4 12 36; do
  for rp in 0 1; do
    python3 tools/predict_prof.py bf16 10 $bs $rp 2>&1 | tail -1
  done
done
python3 tools/predict_prof.py f32 5 4 1 2>&1 | tail -1
python3 tools/predict_prof.py f32 5 4 0 2>&1 | tail -1
