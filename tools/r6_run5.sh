#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6e
rm -rf "$O"; mkdir -p "$O"
cd "$R"
for i in 1 2; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-math --no-pipelined --no-latency --no-fear-m --no-train --dump-ops > "$O/bench_clamp_$i.json" 2> "$O/bench_clamp_$i.err"
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-math --no-pipelined --no-latency --no-fear-m --no-train --no-relu-clamp --dump-ops > "$O/bench_noclamp_$i.json" 2> "$O/bench_noclamp_$i.err"
done
for i in 1 2; do echo "clamp $i"; grep -E "^\s+[0-9]+ \S+\s+[0-9.]+ ms/step|sum of kernels" "$O/bench_clamp_$i.err"; echo "noclamp $i"; grep -E "^\s+[0-9]+ \S+\s+[0-9.]+ ms/step|sum of kernels" "$O/bench_noclamp_$i.err"; done
