#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python bench.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['config4_fear_m_bf16']['value'], d['config5_train_step']['ms_per_step'])"
