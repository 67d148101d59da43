import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from oracle.fear_oracle import OracleNet
from feartracker_amd import DEFAULT_WEIGHTS
net = OracleNet(DEFAULT_WEIGHTS)
print("cpu_count", os.cpu_count())
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    for bs in (8, 32):
        x = torch.randn(bs, 3, 256, 256); z = torch.randn(bs, 256, 8, 8)
        net.track(x, z)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 3: net.track(x, z); n += bs
        print(th, bs, n / (time.perf_counter() - t0), flush=True)
