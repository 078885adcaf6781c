#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CN_FORCE_COMM=1 CN_KEEP_BRANCH_STREAMS=1
rocprofv3 --kernel-trace -d $O/t -o s -- python3 $R/bench.py --dtype bf16 --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $O/line.json 2>/dev/null
python3 $R/tools/timeline.py $O/t/s_results.db 0.5 | head -14
python3 $R/tools/step_trace.py $O/t/s_results.db > $O/step.txt
rm -rf $O/t
tail -1 $O/step.txt
