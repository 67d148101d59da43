#!/bin/bash
# full GPU suite on the current tree
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6g
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -q > "$O/gputests.txt" 2>&1
echo "pytest rc $?" >> "$O/gputests.txt"
grep -n "passed\|failed\|FAILED\|rc \|Error" "$O/gputests.txt" | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
