#!/bin/bash
# round 5: A/B of the row-wise bilinear adjoint (CN_BILINEAR_ROW) inside the step, both precisions; kernel tests first
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_ab2
mkdir -p $O
cd $R
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -q -k "bilinear" 2>&1 | tail -2
run() {
  name=$1; shift
  for P in f32 bf16; do
    A=""; [ $P = bf16 ] && A="--dtype bf16"
    for i in 1 2; do
      env "$@" timeout 300 python3 bench.py $A --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/${name}_${P}_$i.json 2> $O/${name}_${P}_$i.err
      python3 -c "
import json; d=json.load(open('$O/${name}_${P}_$i.json')); print('$name', '$P', round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('kernel_launches_per_step'))"
    done
  done
}
run row X=1
run old CN_BILINEAR_ROW=0
run row2 X=1
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export CN_BILINEAR_ROW=$v
  rocprofv3 --kernel-trace --stats -d $O/stats_$v -o s -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 $R/tools/prof_db.py $O/stats_$v/s_results.db 400 --csv > $O/f32_kernel_stats_row$v.csv
  grep -i 'bilinear' $O/f32_kernel_stats_row$v.csv | head -5
  rm -rf $O/stats_$v
done
