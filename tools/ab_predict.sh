R=$GRAFT_REPO_ROOT
for round in 1 2; do for v in 0 infer; do
CN_PRETIME_FUSED=$v python3 $R/bench.py --child predict --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('fused=$v', 'scene bf16 Mpx/s', round(d['bf16_mixed']['value']/1e6,2), 'fp32', round(d['fp32']['value']/1e6,2), 'tile ms bf16', round(d['tile']['bf16_mixed']['ms_per_tile'],3), 'fp32', round(d['tile']['fp32']['ms_per_tile'],3), 'b4packed', round(d['bf16_mixed']['batch4']['packed']['value']/1e6,2))"
done; done
