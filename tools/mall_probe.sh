#!/bin/bash
# per-op times of the throughput plan with the 256-crop step run as internal passes of 256 / 128 crops (same workspace addresses per pass):
# does the trunk's front (the 128 x 128 and 64 x 64 maps, 268 + 100 MB per 256 crops) gain from staying in the 256 MB Infinity Cache?
cd ${GRAFT_REPO_ROOT:-.}
for mb in 256 128 256 128; do
echo "max-batch $mb"
python bench.py --steps 20 --warmup 5 --max-batch $mb --no-cpu-baseline --no-other-math --no-pipelined --no-latency --no-fear-m --no-train --dump-ops 2>&1 >/dev/null | grep -E "ms/step|sum of"
done
