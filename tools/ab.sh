#!/bin/bash
# A/B runs of the bench on ONE box: each "NAME ENV..." line = one configuration, both precisions, 2 repeats
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-ab}
mkdir -p $O
cd $R
run() {
  name=$1; shift
  for P in f32 bf16; do
    A=""; [ $P = bf16 ] && A="--dtype bf16"
    for i in 1 2; do
      env "$@" timeout 300 python3 bench.py $A --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/${name}_${P}_$i.json 2> $O/${name}_${P}_$i.err
      python3 -c "
import json; d=json.load(open('$O/${name}_${P}_$i.json')); print('$name', '$P', round(d['value'],1), round(d['ms_per_step'],2))"
    done
  done
}
run base X=1
run nohalf CN_BCONV_NO_HALF=1
run side1g CN_SIDE_MAX_WORK=1e9
run side4g CN_SIDE_MAX_WORK=4e9
run side20g CN_SIDE_MAX_WORK=2e10
run noside CN_OVERLAP_WGRAD=0
run base2 X=1
