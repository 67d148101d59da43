#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6h
rm -rf "$O"; mkdir -p "$O"
cd "$R"
{ for b in tools/kb_e1p_perm1 tools/kb_e1p_perm0 tools/kb_e1p_perm1 tools/kb_e1p_perm0; do echo "== $b"; timeout 120 $b 256 30 | grep -v "workgroup  [248]\|workgroup 16"; done; } > "$O/e1pair_perm.txt" 2>&1
{ for a in 0 16 0 16 0 16; do echo "== HC_ABL=$a (16 = __syncthreads() at the end of a layer's last pass)"; timeout 120 tools/kb_hc2_abl$a 256 20 | tail -4; done; } > "$O/headchain_barrier.txt" 2>&1
cat "$O/e1pair_perm.txt" "$O/headchain_barrier.txt"
