#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "chain or plan or hip_net" 2>&1 | tail -2
bash tools/ab_libs.sh
