#!/bin/bash
# Round evidence on the GPU box: bench JSON lines, rocprofv3 kernel stats (as shipped + isolated), queue timelines and
# PMC passes (MFMA busy incl. GRBM_GUI_ACTIVE for the shader clock; FETCH / WRITE traffic) for both precisions.
# usage (through gpurun): bash tools/evidence.sh r03_v1 [nopmc]   -> files under gpurun_out/evidence/
set -u
TAG=${1:-r03}
NOPMC=${2:-}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${TAG}_bench_default.json 2> $O/bench_default.err
python3 $R/bench.py --dtype bf16 > $O/${TAG}_bench_bf16_b32.json 2> $O/bench_bf16.err
# hidden 64 = the reference CLI's default width (scripts/args.yml:220-226), both precisions, short runs
python3 $R/bench.py --hidden 64 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/${TAG}_bench_f32_b8_h64.json 2> $O/bench_h64.err
python3 $R/bench.py --hidden 64 --dtype bf16 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/${TAG}_bench_bf16_b16_h64.json 2>> $O/bench_h64.err
# one-rank RCCL group (the only RCCL path one GPU can exercise): buckets per step and exposed communication time
# (fp32 batch 8 and bf16 batch 32 under a one-rank RCCL group are the `ddp1` block of the default line since round 5)
CN_FORCE_COMM=1 python3 $R/bench.py --hidden 64 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/${TAG}_bench_forcecomm_f32_b8_h64.json 2> $O/bench_fc.err
# ordered kernel traces of one step (hardware queue per kernel): fp32 batch 8 and bf16 batch 32
for P in f32 bf16; do
  A=""; [ $P = bf16 ] && A="--dtype bf16"
  rocprofv3 --kernel-trace -d $O/trace_$P -o s -- python3 $R/bench.py $A --steps 6 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 $R/tools/step_trace.py $O/trace_$P/s_results.db > $O/${TAG}_step_trace_$P.txt
  rm -rf $O/trace_$P
done
# sliding-window predict (36 windows per batch, bf16-mixed): kernel stats + timeline
python3 $R/tools/predict_prof.py bf16 20 36 > $O/${TAG}_predict_bf16_timing.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/stats_pred -o p -- python3 $R/tools/predict_prof.py bf16 10 36 > /dev/null 2>&1
python3 $R/tools/prof_db.py $O/stats_pred/p_results.db 400 --csv > $O/${TAG}_predict_bf16_kernel_stats.csv
python3 $R/tools/timeline.py $O/stats_pred/p_results.db 0.5 > $O/${TAG}_predict_bf16_timeline.txt
rm -rf $O/stats_pred
for P in f32 bf16; do
  A=""; [ $P = bf16 ] && A="--dtype bf16"
  rocprofv3 --kernel-trace --stats -d $O/stats_$P -o s -- python3 $R/bench.py $A --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 $R/tools/prof_db.py $O/stats_$P/s_results.db 400 --csv > $O/${TAG}_bench_${P}_kernel_stats.csv
  python3 $R/tools/timeline.py $O/stats_$P/s_results.db 0.5 > $O/${TAG}_timeline_${P}.txt
  # the same command with the weight-gradient side stream off: every kernel alone on the GPU (bench.py's `isolated`)
  export CN_OVERLAP_WGRAD=0
  rocprofv3 --kernel-trace --stats -d $O/stats_iso_$P -o s -- python3 $R/bench.py $A --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 $R/tools/prof_db.py $O/stats_iso_$P/s_results.db 400 --csv > $O/${TAG}_bench_${P}_kernel_stats_isolated.csv
  if [ -z "$NOPMC" ]; then
  # PMC passes also run serialized, so that a kernel's counters are its own
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
     --kernel-trace --output-format csv -d $O/pmc_mfma_$P -o m -- python3 $R/bench.py $A --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 $R/tools/pmc_mfma.py $O/pmc_mfma_$P/m_counter_collection.csv $O/${TAG}_pmc_mfma_$P.json > $O/${TAG}_pmc_mfma_${P}_summary.txt
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$P -o f -- python3 $R/bench.py $A --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$P -o w -- python3 $R/bench.py $A --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 $R/tools/pmc_traffic.py $O/pmc_fetch_$P/f_counter_collection.csv $O/pmc_write_$P/w_counter_collection.csv $O/${TAG}_pmc_traffic_$P.json > $O/${TAG}_pmc_traffic_${P}_summary.txt
  fi
  unset CN_OVERLAP_WGRAD
  rm -rf $O/stats_$P $O/stats_iso_$P $O/pmc_mfma_$P $O/pmc_fetch_$P $O/pmc_write_$P
done
ls -la $O
