#!/bin/bash
# round 5: the wide-wave bf16 conv (NPW = 2) -- kernel tests, micro-benchmark A/B, step A/B on one box
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_wide
mkdir -p $O
cd $R
timeout 600 python3 -m pytest "tests/test_replay_train_gpu.py::test_plan_recorded_without_a_weight_update_still_repacks" -q -x > $O/pytest_fix1.log 2>&1
echo "pytest rc $?" >> $O/pytest_fix1.log
grep -E "^E |Error|passed|failed|rc " $O/pytest_fix1.log | head -20
timeout 900 python3 -m pytest tests/test_ddp_engine_gpu.py -q -x > $O/pytest_fix2.log 2>&1
echo "pytest rc $?" >> $O/pytest_fix2.log
grep -E "^E |Error|passed|failed|rc " $O/pytest_fix2.log | head -30
timeout 900 python3 -m pytest tests/test_bf16_kernels_gpu.py -q -x -k "conv" > $O/pytest_conv.log 2>&1
echo "pytest rc $?" >> $O/pytest_conv.log
tail -4 $O/pytest_conv.log
for shape in "32 128 100 100 128 3" "32 480 100 100 128 3" "36 128 110 110 128 3" "8 128 200 200 128 3"; do
  for wv in 0 1; do
    echo -n "WIDE=$wv stats "; CN_BCONV_WIDE=$wv timeout 120 python3 tools/bconv_bench.py conv $shape 30 2>&1 | tail -1
    echo -n "WIDE=$wv nostats "; STATS=0 CN_BCONV_WIDE=$wv timeout 120 python3 tools/bconv_bench.py conv $shape 30 2>&1 | tail -1
  done
done
run() {
  name=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python3 bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/${name}_bf16_$i.json 2> $O/${name}_bf16_$i.err
    python3 -c "
import json; d=json.load(open('$O/${name}_bf16_$i.json')); print('$name', 'bf16', round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('kernel_launches_per_step'))"
  done
}
run wide X=1
run narrow CN_BCONV_WIDE=0
run wide_noflush CN_SUM_FLUSH_MB=0
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/f32_flush.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/f32_flush.json')); print('f32 flush128', round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('kernel_launches_per_step'))"
CN_SUM_FLUSH_MB=0 timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/f32_noflush.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/f32_noflush.json')); print('f32 flush0', round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('kernel_launches_per_step'))"
timeout 300 python3 tools/predict_prof.py bf16 20 36 2>&1 | tail -1
CN_BCONV_WIDE=0 timeout 300 python3 tools/predict_prof.py bf16 20 36 2>&1 | tail -1
