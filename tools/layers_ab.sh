#!/bin/bash
# fp32 per-launch contraction tables under two values of an environment switch: bash tools/layers_ab.sh VAR A B [tag]
set -u
VAR=$1; A=$2; B=$3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${4:-layers_ab}
mkdir -p $O
cd $R
for v in $A $B; do
  rm -f /tmp/d.tsv
  env $VAR=$v CN_OVERLAP_WGRAD=0 CN_PROF_DUMP=/tmp/d.tsv timeout 300 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 tools/layerprof.py /tmp/d.tsv 6 > $O/layers_f32_${VAR}_$v.txt
  head -3 $O/layers_f32_${VAR}_$v.txt
done
