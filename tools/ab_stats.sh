#!/bin/bash
# Same-box rocprofv3 kernel stats of the fp32 bench under two settings of one environment switch, every kernel ALONE
# (CN_OVERLAP_WGRAD=0): bash tools/ab_stats.sh VAR A B [tag] -> gpurun_out/<tag>/stats_<VAR>_<value>.csv
VAR=$1; A=$2; B=$3; TAG=${4:-abstats}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CN_OVERLAP_WGRAD=0
for v in $A $B; do
  export $VAR=$v
  rocprofv3 --kernel-trace --stats -d $O/st_$v -o s -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 $R/tools/prof_db.py $O/st_$v/s_results.db 400 --csv > $O/stats_${VAR}_$v.csv
  rm -rf $O/st_$v
done
