#!/bin/bash
# Collects everything profiles/rNN_* is built from, on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh'
# then fold the results locally with tools/profile_fold.sh <round>.
#   1. python bench.py                       -> gpurun_out/round/bench.json (+ cpu_baseline)
#   2. rocprofv3 --kernel-trace --stats       (math 0 and math 1, --dump-ops: per-op table on stderr)
#   3. rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (no other trace domains)
#   4. the same for configs[3] (tools/fear_m_prof.py); configs[4] (tools/train_prof.py): kernel stats, unprofiled timing, PMC traffic
#   5. SQ counters of the hot path (three --pmc passes)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python "$R/bench.py" > "$O/bench.json" 2> "$O/bench.err"
for m in 0 1; do
  rocprofv3 --kernel-trace --stats -d "$O/trace_math$m" -o p --output-format csv -- \
    python "$R/bench.py" --steps 20 --warmup 5 --math $m --no-cpu-baseline --no-other-math --no-pipelined --no-latency --no-fear-m --no-train --dump-ops > "$O/trace_math$m.json" 2> "$O/trace_math$m.err"
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d "$O/pmc_$c" -o p --output-format csv -- \
    python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-other-math --no-pipelined --no-latency --no-fear-m --no-train > "$O/pmc_$c.json" 2> "$O/pmc_$c.err"
done
# BASELINE configs[3] (synthetic FEAR-M, bf16, B=512): kernel trace + the two PMC passes of the same script
rocprofv3 --kernel-trace --stats -d "$O/fear_m_trace" -o p --output-format csv -- python "$R/tools/fear_m_prof.py" 10 > "$O/fear_m_trace.out" 2> "$O/fear_m_trace.err"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d "$O/fear_m_pmc_$c" -o p --output-format csv -- python "$R/tools/fear_m_prof.py" 3 > "$O/fear_m_pmc_$c.out" 2> "$O/fear_m_pmc_$c.err"
done
# BASELINE configs[4] (one rank's 128 pairs of the training step): kernel stats
# (the default, block-fused implementation; 1 warm-up + 5 timed + 1 phase-marked step = 7 steps in the trace) and the two PMC passes of
# the same script for tools/train_traffic.py (1 + 2 + 1 = 4 steps each)
rocprofv3 --kernel-trace --stats -d "$O/train_trace" -o p --output-format csv -- python "$R/tools/train_prof.py" 128 5 block > "$O/train_trace.out" 2> "$O/train_trace.err"
python "$R/tools/train_prof.py" 128 8 block > "$O/train_plain.out" 2> "$O/train_plain.err"      # the same step without the profiler
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d "$O/train_pmc_$c" -o p --output-format csv -- python "$R/tools/train_prof.py" 128 2 block > "$O/train_pmc_$c.out" 2> "$O/train_pmc_$c.err"
done
# SQ counters of the hot path's kernels (tools/pmc_summary.py -> profiles/rNN_sq_counters.txt)
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d "$O/sq_pass$i" -o p --output-format csv -- \
    python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-other-math --no-pipelined --no-latency --no-fear-m --no-train > "$O/sq_pass$i.json" 2> "$O/sq_pass$i.err"
done
# ... and of the training step's (tools/pmc_summary.py --all -> profiles/rNN_train_sq_counters.txt)
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d "$O/train_sq_pass$i" -o p --output-format csv -- python "$R/tools/train_prof.py" 128 1 block > "$O/train_sq_pass$i.out" 2> "$O/train_sq_pass$i.err"
done
# the distributed code path (RCCL process group, barriers, fear_track_packed + all-gather) with the one rank a 1-GPU box has
FEAR_BENCH_FORCE_DIST=1 python "$R/bench.py" --no-cpu-baseline --no-other-math > "$O/bench_force_dist.json" 2> "$O/bench_force_dist.err"
tail -c 600 "$O/bench.json"
