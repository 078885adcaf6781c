#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_quick2}
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_bf16_kernels_gpu.py tests/test_bf16_model_gpu.py tests/test_predict_edges_gpu.py -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "^FAILED|passed|failed" $O/pytest.log | tail -8
python3 tools/predict_prof.py bf16 10 2>&1 | tail -1
CN_EVAL_FUSION=0 python3 tools/predict_prof.py bf16 10 2>&1 | tail -1
