#!/bin/bash
# the data-parallel stream set with / without the auxiliary branch streams under a one-rank RCCL group,
# with the engine's streams created before / after the process group, at 4 and 8 hardware queues. bf16 batch 32 + fp32 batch 8.
R=$GRAFT_REPO_ROOT
run() {
  name=$1; shift
  for P in bf16 f32; do
    X=""; [ $P = bf16 ] && X="--dtype bf16"
    env CN_FORCE_COMM=1 "$@" python3 $R/bench.py $X --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$name $P', round(d['value'],1), round(d['ms_per_step'],3), d['config'].get('streams_created_before_process_group'))"
  done
}
run no_branches X=1
run branches CN_KEEP_BRANCH_STREAMS=1
run branches_streams_first CN_KEEP_BRANCH_STREAMS=1 CN_STREAMS_FIRST=1
run no_branches_streams_first CN_STREAMS_FIRST=1
run branches_streams_first_q4 CN_KEEP_BRANCH_STREAMS=1 CN_STREAMS_FIRST=1 GPU_MAX_HW_QUEUES=4
run branches_q4 CN_KEEP_BRANCH_STREAMS=1 GPU_MAX_HW_QUEUES=4
run branches_streams_first_auxnormal CN_KEEP_BRANCH_STREAMS=1 CN_STREAMS_FIRST=1 CN_AUX_STREAM=normal
run single_process_reference X=1 CN_FORCE_COMM=0
