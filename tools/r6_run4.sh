#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6d
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "relu_clamp or e1_pair or golden or oracle" > "$O/gputests_clamp.txt" 2>&1
echo "pytest rc $?" >> "$O/gputests_clamp.txt"
grep -n "passed\|failed\|FAILED\|rc \|Error" "$O/gputests_clamp.txt" | head
for i in 1 2 3; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-math --no-pipelined --no-latency --no-fear-m --no-train > "$O/bench_clamp_$i.json" 2> "$O/bench_clamp_$i.err"
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-math --no-pipelined --no-latency --no-fear-m --no-train --no-relu-clamp > "$O/bench_noclamp_$i.json" 2> "$O/bench_noclamp_$i.err"
done
for f in "$O"/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4))
PY
done
