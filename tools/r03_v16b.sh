#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_v16b}
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_bf16_kernels_gpu.py -m gpu -q > $O/pytest_k.log 2>&1
echo "rc $?" >> $O/pytest_k.log
grep -E "^FAILED|passed|failed|^E  " $O/pytest_k.log | head -12
for V in 1 0; do
  rm -f /tmp/d.tsv
  CN_BCONV_V16=$V CN_OVERLAP_WGRAD=0 CN_PROF_DUMP=/tmp/d.tsv timeout 300 python3 bench.py --dtype bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/b_$V.json 2> $O/b_$V.err
  python3 tools/layerprof.py /tmp/d.tsv 3 > $O/layers_v$V.txt
  echo "== v16=$V"; grep "bconv" $O/layers_v$V.txt | head -14
done
