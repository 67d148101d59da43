#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6f
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_train_head.py tests/test_train_syncbn.py -m gpu -x -q -k "graphed or syncbn or whole_network or sync" > "$O/gputests.txt" 2>&1
echo "pytest rc $?" >> "$O/gputests.txt"
grep -n "passed\|failed\|FAILED\|rc \|Error" "$O/gputests.txt" | head
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-math --no-pipelined --no-latency --no-fear-m > "$O/bench_train.json" 2> "$O/bench_train.err"
python - "$O/bench_train.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); t=d["config5_train_step"]
print("value", d["value"])
for k in ("ms_per_step","host_issue_ms_per_step","graphed_step","sync_bn_one_rank_ms"): print(k, t.get(k))
PY
tail -3 "$O/bench_train.err"
