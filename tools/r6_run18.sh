#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "chain or plan or hip_net or clip or update" 2>&1 | tail -3
bash tools/r6_run13.sh
