#!/bin/bash
# same-box A/B of the fused PreTimeReduction in training: generic / fused with its backward on the side / compute stream
R=$GRAFT_REPO_ROOT
for round in 1 2; do
  for cfg in "0 main" "1 side" "1 main"; do
    set -- $cfg
    for P in bf16 f32; do
      X=""; [ $P = bf16 ] && X="--dtype bf16"
      CN_PRETIME_FUSED=$1 CN_PRETIME_BWD=$2 python3 $R/bench.py $X --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('fused=$1 bwd=$2 $P', round(d['value'],1), round(d['ms_per_step'],3), d['config']['kernel_launches_per_step'])"
    done
  done
done
