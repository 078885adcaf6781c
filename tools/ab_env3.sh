#!/bin/bash
# same-box A/B of an environment switch: bf16 / fp32 train, reference-default point, predict; 3 interleaved rounds
VAR=$1; A=$2; B=$3
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in $A $B; do
    for P in bf16 f32; do
      X=""; [ $P = bf16 ] && X="--dtype bf16"
      env $VAR=$v python3 $R/bench.py $X --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$VAR=$v $P', round(d['value'],1), round(d['ms_per_step'],3), d['config']['kernel_launches_per_step'])"
    done
    env $VAR=$v python3 $R/bench.py --child default_point --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$VAR=$v default_point', round(d['value'],1), round(d['ms_per_step'],3))"
    env $VAR=$v python3 $R/tools/predict_prof.py bf16 20 36 2>/dev/null | grep Mpx | sed "s/^/$VAR=$v /"
  done
done
