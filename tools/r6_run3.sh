#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests/test_train_syncbn.py -m gpu -q 2>&1 | tail -3
{ for b in tools/kb_ir16_gs2_d4 tools/kb_ir16_gs3_d3 tools/kb_ir16_gs3_d6 tools/kb_ir16_gs5_d5 tools/kb_ir16_gs5_d10 tools/kb_ir16_gs6_d6 tools/kb_ir16_gs6_d12 tools/kb_ir16_gs10_d10 tools/kb_ir16_gs2_d4; do echo "== $b"; timeout 120 $b 256 20 | grep "fp32-mfma\|IR16"; done; } > "$O/ir16_gs_sweep.txt" 2>&1
cat "$O/ir16_gs_sweep.txt"
