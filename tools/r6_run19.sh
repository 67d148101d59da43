#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do for l in libfear_hip_prev.so libfear_hip.so; do
FEAR_LIB=feartracker_amd/$l python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --no-other-math --no-fear-m --no-train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['latency_batch1']; print('$l', 'value', round(d['value']), 'b1 track ms', round(l['track_call_batch1_ms'],4), 'update', {k:(round(v['total'],4) if isinstance(v,dict) and 'total' in v else None) for k,v in l.items() if isinstance(v,dict)})"
done; done
