#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_slots; mkdir -p $O
run() {
  name=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/${name}_$i.json 2> $O/${name}_$i.err
    python3 -c "
import json; d=json.load(open('$O/${name}_$i.json')); print('$name', 'f32', round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('kernel_launches_per_step'))"
  done
}
run base X=1
run slots512 CN_WGX_SLOTS=512
run slots1024 CN_WGX_SLOTS=1024
run slots2048 CN_WGX_SLOTS=2048
run nbuf1 CN_WGX_NBUF=1
run base2 X=1
