#!/bin/bash
# Round 6: how many hardware queues the training step tolerates with and without a one-rank RCCL group (CN_FORCE_COMM=1).
# Every configuration has the weight-gradient stream (lowest priority) and ONE auxiliary stream (highest priority);
# GPU_MAX_HW_QUEUES caps the normal-priority pool. Result of the run that fixed the shipped setting: profiles/r06_hw_queues.txt
R=$GRAFT_REPO_ROOT
run() {
  name=$1; shift
  for P in bf16 f32; do
    X=""; [ $P = bf16 ] && X="--dtype bf16"
    env "$@" python3 $R/bench.py $X --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$name $P', round(d['value'],1), round(d['ms_per_step'],3))"
  done
}
run single_q5 X=1
run single_q8 GPU_MAX_HW_QUEUES=8
run ddp1_q5 CN_FORCE_COMM=1
run ddp1_q4 CN_FORCE_COMM=1 GPU_MAX_HW_QUEUES=4
run ddp1_q6_aux_forced CN_FORCE_COMM=1 CN_KEEP_BRANCH_STREAMS=1 GPU_MAX_HW_QUEUES=6
run ddp1_q8_no_aux CN_FORCE_COMM=1 GPU_MAX_HW_QUEUES=8
run ddp1_q5_no_aux CN_FORCE_COMM=1 CN_HEAD_STREAMS=0
