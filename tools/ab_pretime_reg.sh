#!/bin/bash
# same-box A/B of the register variant of the PreTimeReduction kernels (CN_PRETIME_REG=0: the generic kernel)
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests/test_pretime_gpu.py -x -q 2>&1 | tail -3
for round in 1 2; do
  for reg in 0 1; do
    for B in 32 8; do
      for kind in 1 0; do
        echo "reg=$reg B=$B kind=$kind: $(CN_PRETIME_REG=$reg python3 tools/pretime_bench.py $B 3 12 100 32 $kind 2>&1 | awk '{printf "%s %s %s | ", $1, $2, $3}')"
      done
    done
  done
done
