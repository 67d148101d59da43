#!/bin/bash
# round 6, GPU call: SyncBatchNorm hook tests + the training tests, the training objects of the bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6b
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 1200 python -m pytest tests/test_train_syncbn.py tests/test_train_block.py tests/test_train_head.py tests/test_train_optim.py tests/test_abi.py tests/test_export.py -m gpu -q > "$O/gputests_train.txt" 2>&1
echo "pytest rc $?" >> "$O/gputests_train.txt"
grep -n "passed\|failed\|FAILED\|rc " "$O/gputests_train.txt"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-math --no-pipelined --no-latency --no-fear-m > "$O/bench_train.json" 2> "$O/bench_train.err"
wc -l "$O/bench_train.json"
