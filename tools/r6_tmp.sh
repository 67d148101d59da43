#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
FEAR_LIB=feartracker_amd/libfear_hip_debug.so timeout 2000 python -X faulthandler -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Fatal|File \"/root/repo/tests" | tail -8
