#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "mid_size or chain or plan or ragged or toggles" 2>&1 | tail -3
python - <<'PY'
import torch, time
from feartracker_amd import FEARNetHIP
from tests.conftest import WEIGHTS
for nb in (104, 128, 143, 144, 160, 256):
    n = FEARNetHIP(WEIGHTS, device=0, max_batch=nb)
    x = torch.randn(nb,3,256,256,device='cuda'); z = n.get_features(torch.randn(nb,3,128,128,device='cuda'))
    for _ in range(5): n.track_maps(x,z)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(30): n.track_maps(x,z)
    torch.cuda.synchronize(); dt=(time.time()-t)/30
    print('crops', nb, 'default ms', round(dt*1e3,4), [nm for nm,_,_ in n.plan(256, True)][3:5], flush=True)
PY
