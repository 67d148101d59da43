#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "chain32" 2>&1 | tail -15
python - <<'PY'
import torch, time
from feartracker_amd import FEARNetHIP
from tests.conftest import WEIGHTS
for on in (True, False, True, False):
    n = FEARNetHIP(WEIGHTS, device=0, max_batch=256); n.set_small_pass(0); n.set_chain32(on)
    x = torch.randn(256,3,256,256,device='cuda'); z = n.get_features(torch.randn(256,3,128,128,device='cuda'))
    for _ in range(5): n.track_maps(x,z)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(20): n.track_maps(x,z)
    torch.cuda.synchronize(); dt=(time.time()-t)/20
    print('chain32', on, 'ms/256', round(dt*1e3,4), 'crops/s', round(256/dt))
PY
