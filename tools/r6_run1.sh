#!/bin/bash
# round 6, GPU call 1: kbench A/B of the e1pair rewrite, headchain scratch ablations, the GPU suite, a short bench line with the per-op table
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6a
rm -rf "$O"; mkdir -p "$O"
cd "$R"
{
  for b in tools/kb_e1p_s8_na3 tools/kb_e1p_s8_na5 tools/kb_e1p_s0_na3 tools/kb_e1p_s8_na1; do echo "== $b"; timeout 120 $b 256 30; done
} > "$O/e1pair_kbench.txt" 2>&1
{
  for a in 0 1 2 3 4 0; do echo "== HC_ABL=$a"; timeout 120 tools/kb_hc_abl$a 256 20 | tail -4; done
} > "$O/headchain_abl.txt" 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/gputests.txt" 2>&1
echo "pytest rc $?" >> "$O/gputests.txt"
for i in 1 2; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-math --no-pipelined --no-latency --no-fear-m --no-train --dump-ops > "$O/bench_e1pair_$i.json" 2> "$O/bench_e1pair_$i.err"
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-math --no-pipelined --no-latency --no-fear-m --no-train --no-e1-pair > "$O/bench_noe1pair_$i.json" 2> "$O/bench_noe1pair_$i.err"
done
tail -3 "$O/gputests.txt"; cat "$O/e1pair_kbench.txt" "$O/headchain_abl.txt"; for f in "$O"/bench_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
