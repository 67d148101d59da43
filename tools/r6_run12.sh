#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6m
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fear_m or matrix_pipe or bf16 or math or split_mode" > "$O/gputests.txt" 2>&1
echo "pytest rc $?" >> "$O/gputests.txt"
grep -n "passed\|failed\|FAILED\|rc \|Error" "$O/gputests.txt" | head
python tools/fear_m_prof.py 10 2 512 2>&1 | grep -E "ir16_|sum of kernels|crops/s|ms" | head -24
