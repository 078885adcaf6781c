#!/bin/bash
# same-box A/B of the register variant of the PreTimeReduction kernels (CN_PRETIME_REG=0: the generic kernel),
# with the per-pass kernel times of each build (the library's own HIP-event profiler)
R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
python3 -m pytest tests/test_pretime_gpu.py -x -q 2>&1 | tail -1
for LIB in ${LIBS:-libcultionet_hip.so}; do
  export CN_LIB_PATH=$R/cultionet_amd/csrc/$LIB
  for reg in ${REGS:-1}; do
    for B in 32 8; do
      echo "$LIB reg=$reg B=$B: $(CN_PRETIME_REG=$reg python3 tools/pretime_bench.py $B 3 12 100 32 1 2>&1 | grep "us \|passes" | awk '/passes/ {print ""; print; next} {printf "%s %s %s | ", $1, $2, $3}')"
    done
  done
done
