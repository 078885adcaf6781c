#!/bin/bash
# same-box A/B of register-variant builds: training passes at batch 32 / 8 and the scene predictor's cube (C = 4, T = 25)
R=$GRAFT_REPO_ROOT
cd $R
for round in 1 2; do
  for LIB in ${LIBS:-libcultionet_hip.so}; do
    export CN_LIB_PATH=$R/cultionet_amd/csrc/$LIB
    echo "$LIB: $(python3 tools/pretime_bench.py 36 4 25 110 32 1 2>&1 | grep 'eval fwd')"
    python3 tools/pretime_bench.py 32 3 12 100 32 1 2>&1 | grep "passes"
    python3 tools/pretime_bench.py 8 3 12 100 32 0 2>&1 | grep "passes"
  done
done
