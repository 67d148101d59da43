"""BASELINE configs[3] (synthetic FEAR-M, bf16 matrix-pipe mode, B=512) for rocprofv3: prints the launch plan's per-op table
(HIP events, `--dump-ops` format of bench.py) to stderr, then runs `steps` more passes — the last fear:: dispatches of the
trace, which tools/trace_to_ops.py folds by plan position.
    rocprofv3 --kernel-trace --stats ... -- python tools/fear_m_prof.py 10        (and --pmc FETCH_SIZE / WRITE_SIZE passes)
usage: fear_m_prof.py [steps=10] [math=2] [batch=512]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from feartracker_amd import FEARNetHIP
from feartracker_amd.hip_backend import WEIGHTS_FEAR_M

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
math = int(sys.argv[2]) if len(sys.argv) > 2 else 2
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dev = torch.device("cuda:0")
net = FEARNetHIP(WEIGHTS_FEAR_M, device=0, max_batch=batch)
net.set_math(math)
g = torch.Generator().manual_seed(4)
mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1) * 255.0
inv = 1.0 / (torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1) * 255.0)
x = ((torch.randint(0, 256, (batch, 3, 256, 256), dtype=torch.uint8, generator=g).float() - mean) * inv).to(dev).contiguous()
t = ((torch.randint(0, 256, (batch, 3, 128, 128), dtype=torch.uint8, generator=g).float() - mean) * inv).to(dev).contiguous()
z = net.get_features(t)
bbox = torch.empty((batch, 4, 16, 16), device=dev)
cls = torch.empty((batch, 1, 16, 16), device=dev)
for _ in range(3):
    net.track_maps(x, z, out=(bbox, cls))
torch.cuda.synchronize()
plan = net.plan(256, True)
net.set_profile(True, op=-1)
net.profile_reset()
for _ in range(3):
    net.track_maps(x, z, out=(bbox, cls))
torch.cuda.synchronize()
prof = net.profile_read(256, True)
net.set_profile(False)
tot = sum(ms / 3 for ms, _ in prof)
for i, ((name, fl, by), (ms, cnt)) in enumerate(zip(plan, prof)):
    per = ms / max(cnt, 1)
    cpl = batch / max(cnt / 3, 1)
    print(f"{i:3d} {name:28s} {ms / 3:8.3f} ms/step {100 * ms / 3 / tot:5.1f}%  {fl * cpl / (per * 1e-3) / 1e12 if per else 0:7.2f} TF/s "
          f"{by * cpl / (per * 1e-3) / 1e9 if per else 0:8.1f} GB/s", file=sys.stderr)
print(f"sum of kernels {tot:.3f} ms/step", file=sys.stderr)
for _ in range(steps):
    net.track_maps(x, z, out=(bbox, cls))
torch.cuda.synchronize()
