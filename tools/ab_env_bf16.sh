#!/bin/bash
# same-box A/B of an environment switch on the bf16 step only, N interleaved rounds: bash tools/ab_env_bf16.sh VAR A B [rounds]
VAR=$1; A=$2; B=$3; N=${4:-4}
R=$GRAFT_REPO_ROOT
for round in $(seq 1 $N); do
  for v in $A $B; do
    env $VAR=$v python3 $R/bench.py --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$VAR=$v', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
