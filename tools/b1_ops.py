"""Batch-1 `track` call: ms per call (calls back to back on one stream) and the per-op table of the one-crop launch plan
(HIP events around every op: each op carries a few microseconds of event overhead).  -> profiles/r03_batch1_ops.txt"""
import os
import sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feartracker_amd import FEARNetHIP, DEFAULT_WEIGHTS
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 3, 256, 256, generator=g).cuda(); t = torch.randn(1, 3, 128, 128, generator=g).cuda()
net = FEARNetHIP(DEFAULT_WEIGHTS, device=0, max_batch=1)
z = net.get_features(t)
for rep in range(2):
    for _ in range(50): net.track_maps(x, z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): net.track_maps(x, z)
    torch.cuda.synchronize(); print("ms/track back-to-back %.4f" % ((time.perf_counter() - t0) / 500 * 1e3))
net.set_plan_crops(1)
plan = net.plan(256, True)
net.set_profile(True)
for _ in range(100): net.track_maps(x, z)
torch.cuda.synchronize()
pr = net.profile_read(256, True)
tot = 0
for op, (ms, n) in zip(plan, pr):
    print("%-40s %.1f us" % (op[0], ms / max(n, 1) * 1e3)); tot += ms / max(n, 1)
print("sum us", tot * 1e3, "ops", len(plan))
