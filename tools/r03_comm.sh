#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd $R
for M in low normal; do
 for C in 1 0; do
  for i in 1 2; do
   CN_SIDE_STREAM=$M CN_FORCE_COMM=$C timeout 300 python3 bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('side=$M comm=$C bf16', round(d['value'],1), round(d['ms_per_step'],2), d['config']['comm_ms_exposed'])"
  done
 done
done
