#!/bin/bash
# A/B of two builds of the library on ONE box: feartracker_amd/libfear_hip_prev.so (copy the old build there) against libfear_hip.so,
# alternating, two rounds: fp32 value, FEAR-M bf16, fp16-split.   gpurun --timeout 900 -- bash tools/ab_libs.sh
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do for l in libfear_hip_prev.so libfear_hip.so; do
FEAR_LIB=feartracker_amd/$l python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined --no-latency --no-train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', 'value', round(d['value']), 'fear_m bf16', round(d['config4_fear_m_bf16']['value']), 'fp16split', round(d['other_math_mode']['value']))"
done; done
