#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03_pred}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/predict_prof.py bf16 10 > $O/plain.txt 2>&1
cat $O/plain.txt | tail -1
rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python3 $R/tools/predict_prof.py bf16 5 > $O/prof.txt 2>&1
tail -1 $O/prof.txt
python3 $R/tools/prof_db.py $O/prof/p_results.db 45 > $O/kernels.txt 2> $O/kernels.err
cat $O/kernels.err; head -45 $O/kernels.txt
rm -rf $O/prof
