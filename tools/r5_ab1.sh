#!/bin/bash
# round 5, call 1: GPU suite (with the new gradient-probe printouts, deferred slice sums) + A/B of CN_WGRAD_CUS and
# CN_DEFER_SUMS on one box
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_ab1
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -s --durations=8 > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed|rc |grad probes|batch 32|bitwise" $O/pytest.log | tail -40
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
run base X=1
run nodefer CN_DEFER_SUMS=0
run cus224 CN_WGRAD_CUS=224
run cus192 CN_WGRAD_CUS=192
run cus160 CN_WGRAD_CUS=160
run base2 X=1
