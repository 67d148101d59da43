#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do for l in libfear_hip.so libfear_var_gs3d6.so libfear_var_gs2d6.so libfear_var_gs5d5.so libfear_var_d4_5.so libfear_var_d2_5.so; do
FEAR_LIB=feartracker_amd/$l python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined --no-latency --no-train --no-fear-m --no-other-math 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', 'value', round(d['value']), 'ms', round(d['ms_per_step'],4))"
done; done
