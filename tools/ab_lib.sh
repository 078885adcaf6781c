#!/bin/bash
# Same-box A/B of two builds of the library: build the committed HEAD's csrc to /tmp, ship it next to the working-tree
# build, alternate the two under CN_LIB_PATH.   usage (build container):  bash tools/ab_lib.sh "--dtype bf16" 3
set -u
ARGS=${1:-"--dtype bf16"}
REPS=${2:-3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/csrc_prev && mkdir -p /tmp/csrc_prev
for f in $(git -C $ROOT ls-files cultionet_amd/csrc); do git -C $ROOT show HEAD:$f > /tmp/csrc_prev/$(basename $f); done
make -C /tmp/csrc_prev -j8 > /tmp/csrc_prev/build.log 2>&1 || { tail -5 /tmp/csrc_prev/build.log; exit 1; }
cp /tmp/csrc_prev/libcultionet_hip.so $ROOT/cultionet_amd/csrc/libcultionet_hip_prev.so
/usr/local/graft/bin/gpurun --timeout 1200 -- "for i in \$(seq $REPS); do for L in prev new; do P=''; [ \$L = prev ] && P=cultionet_amd/csrc/libcultionet_hip_prev.so; CN_LIB_PATH=\$P timeout 300 python3 bench.py $ARGS --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c \"
import json,sys; d=json.load(sys.stdin); print('\$L', round(d['value'],1), round(d['ms_per_step'],2))\"; done; done" 2>&1 | grep -E "^(prev|new) "
rm -f $ROOT/cultionet_amd/csrc/libcultionet_hip_prev.so
