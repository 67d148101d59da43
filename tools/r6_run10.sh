#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6k
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q > "$O/gputests.txt" 2>&1
echo "pytest rc $?" >> "$O/gputests.txt"
grep -n "passed\|failed\|FAILED\|rc \|Error" "$O/gputests.txt" | head
for i in 1 2 3; do python tools/train_prof.py 128 8 block 2>&1 | grep "mode="; done
