#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6n
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6n/gputests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6n/gputests.txt
tail -5 gpurun_out/r6n/gputests.txt
for i in 1 2; do for l in libfear_hip_prev.so libfear_hip.so; do
FEAR_LIB=feartracker_amd/$l python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined --no-latency --no-train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', 'value', round(d['value']), 'ms', d['ms_per_step'], 'fear_m bf16', round(d['config4_fear_m_bf16']['value']))"
done; done
