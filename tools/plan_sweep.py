"""ms per fear_track call against the pass size, for the small-batch plans (FEAR_OPT_SMALL_PASS = 96) and the throughput plan (0).
Built with -DFEAR_TINY_PASS=0 / 32 (FEAR_LIB=...) it separates the tiny plan from the small plan: profiles/r03_plan_sweep.txt."""
import sys, time, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feartracker_amd import FEARNetHIP, DEFAULT_WEIGHTS
g = torch.Generator().manual_seed(0)
X = torch.randn(128, 3, 256, 256, generator=g).cuda(); T = torch.randn(128, 3, 128, 128, generator=g).cuda()
tag = os.path.basename(os.environ.get("FEAR_LIB", "default"))
for small in (96, 0):
    net = FEARNetHIP(DEFAULT_WEIGHTS, device=0, max_batch=128)
    net.set_small_pass(small)
    Z = net.get_features(T)
    row = []
    for n in (1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128):
        x, z = X[:n].contiguous(), Z[:n].contiguous()
        for _ in range(10): net.track_maps(x, z)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(60): net.track_maps(x, z)
        torch.cuda.synchronize(); row.append("%d:%.3f" % (n, (time.perf_counter() - t0) / 60 * 1e3))
    print(tag, "small_pass", small, " ".join(row))
