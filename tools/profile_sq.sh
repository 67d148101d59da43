#!/bin/bash
# SQ counters (MFMA / VALU / LDS utilisation, wait classes) of the fear:: kernels, three rocprofv3 --pmc passes of a short
# bench run (no other trace domains):  gpurun --timeout 900 -- 'bash tools/profile_sq.sh'
# then locally:  python tools/pmc_summary.py gpurun_out/sq > profiles/rNN_sq_counters.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sq
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d "$O/pass$i" -o p --output-format csv -- \
    python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-other-math --no-pipelined --no-latency --no-fear-m --no-train > "$O/pass$i.json" 2> "$O/pass$i.err"
done
ls "$O"
