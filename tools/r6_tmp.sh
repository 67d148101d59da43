#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fear_m or matrix_pipe or bf16 or math or split_mode" 2>&1 | tail -2
bash tools/ab_libs.sh
