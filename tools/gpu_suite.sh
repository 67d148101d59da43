#!/bin/bash
# the GPU suite into gpurun_out/r6n/gputests.txt
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6n
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6n/gputests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6n/gputests.txt
tail -4 gpurun_out/r6n/gputests.txt
