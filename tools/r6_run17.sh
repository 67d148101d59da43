#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --steps 20 --warmup 5 --math 0 --no-cpu-baseline --no-other-math --no-pipelined --no-latency --no-fear-m --no-train --dump-ops 2>&1 >/dev/null | grep -v "^$" | tail -20
