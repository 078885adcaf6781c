#!/bin/bash
# per-launch dumps (CN_PROF_DUMP) of the contraction kernels for both precisions, isolated (side stream off)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-layers}
mkdir -p $O
cd $R
for P in f32 bf16; do
  A=""; [ $P = bf16 ] && A="--dtype bf16"
  rm -f /tmp/d_$P.tsv
  CN_OVERLAP_WGRAD=0 CN_PROF_DUMP=/tmp/d_$P.tsv timeout 300 python3 bench.py $A --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/b_$P.json 2> $O/b_$P.err
  python3 tools/layerprof.py /tmp/d_$P.tsv 6 > $O/layers_$P.txt
  head -5 $O/layers_$P.txt
done
