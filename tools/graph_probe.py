"""Is the batch-1 network launch-bound?  Eager launches vs one captured graph replay of the same fear_track call
(development probe; run on the GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feartracker_amd import FEARNetHIP, DEFAULT_WEIGHTS

net = FEARNetHIP(DEFAULT_WEIGHTS, device=0, max_batch=1)
g = torch.Generator().manual_seed(0)
x = torch.randn(1, 3, 256, 256, generator=g).cuda()
z = net.get_features(torch.randn(1, 3, 128, 128, generator=g).cuda())
bbox = torch.empty(1, 4, 16, 16, device="cuda"); cls = torch.empty(1, 1, 16, 16, device="cuda")
for _ in range(20):
    net.track_maps(x, z, out=(bbox, cls))
torch.cuda.synchronize()
ref = (bbox.clone(), cls.clone())

def timeit(fn, n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

def timeit_sync(fn, n=300):
    torch.cuda.synchronize(); t = 0.0
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t += time.perf_counter() - t0
    return 1e3 * t / n

eager = lambda: net.track_maps(x, z, out=(bbox, cls))
print("eager back-to-back ms/call", timeit(eager), " eager sync-each ms/call", timeit_sync(eager))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        eager()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(gr, stream=s):
        eager()
    torch.cuda.synchronize()
    bbox.zero_(); cls.zero_()
    gr.replay(); torch.cuda.synchronize()
    print("graph result identical:", torch.equal(bbox, ref[0]) and torch.equal(cls, ref[1]))
    print("graph back-to-back ms/call", timeit(gr.replay), " graph sync-each ms/call", timeit_sync(gr.replay))
except Exception as e:
    print("capture failed:", repr(e))
# single-stream variant (profile mode keeps one stream): how much do the two streams buy?
net.set_plan_crops(1)
net.set_profile(True, op=-1); net.profile_reset()
for _ in range(50):
    eager()
torch.cuda.synchronize()
plan = net.plan(256, True); prof = net.profile_read(256, True)
net.set_profile(False)
tot = 0.0
for (name, fl, by), (ms, cnt) in zip(plan, prof):
    t = 1e3 * ms / max(cnt, 1); tot += t
    print(f"  {name:34s} {t:7.1f} us  x{cnt}")
print("sum (single stream, profiled)", tot, "us")
