#!/bin/bash
# GPU box: ordered kernel trace of one bf16 step (batch 32 hidden 32, and the reference-default point) under rocprofv3
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-trace}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/t1 -o s -- python3 $R/bench.py --dtype bf16 --steps 6 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 $R/tools/step_trace.py $O/t1/s_results.db > $O/step_bf16_b32.txt
rocprofv3 --kernel-trace -d $O/t2 -o s -- python3 $R/bench.py --dtype bf16 --hidden 64 --batch 4 --steps 6 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 $R/tools/step_trace.py $O/t2/s_results.db > $O/step_bf16_h64_b4.txt
rm -rf $O/t1 $O/t2
tail -1 $O/step_bf16_b32.txt; tail -1 $O/step_bf16_h64_b4.txt
