#!/bin/bash
# same-box A/B of the tower heads spawned beside the tower convolutions (CN_HEAD_STREAMS): train bf16 / fp32, default point, predict
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in 0 1; do
    for P in bf16 f32; do
      X=""; [ $P = bf16 ] && X="--dtype bf16"
      CN_HEAD_STREAMS=$v python3 $R/bench.py $X --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('heads=$v $P', round(d['value'],1), round(d['ms_per_step'],3), d['config']['kernel_launches_per_step'])"
    done
    CN_HEAD_STREAMS=$v python3 $R/bench.py --child default_point --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('heads=$v default_point', round(d['value'],1), round(d['ms_per_step'],3), 'eager', round(d['eager']['value'],1))"
    CN_HEAD_STREAMS=$v python3 $R/tools/predict_prof.py bf16 20 36 2>/dev/null | grep Mpx | sed "s/^/heads=$v /"
  done
done
