"""The 128-pair training step (BASELINE configs[4], one rank's share) a few times, for rocprofv3:
    cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trainprof -o p --output-format csv -- python $R/tools/train_prof.py
(tools/profile_round.sh does it and tools/profile_fold.sh copies p_kernel_stats.csv to profiles/rNN_train_kernel_stats.csv).
usage: train_prof.py [pairs=128] [steps=5] [mode=block|layerwise|fused (or 0 / 1 = layerwise / fused)] [two_streams=1] [virtual_expansion=32]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from feartracker_amd.train_net import FEARNetTrainHIP, random_init_state

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else "block"
mode = {"0": "layerwise", "1": "fused"}.get(mode, mode)
two = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
virt = int(sys.argv[5]) if len(sys.argv) > 5 else 32
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
net = FEARNetTrainHIP(random_init_state(3), device=0, mode=mode, two_streams=two, virtual_expansion=virt)
tmpl = torch.randn(batch, 3, 128, 128, generator=g).to(dev)
srch = torch.randn(batch, 3, 256, 256, generator=g).to(dev)
gt_reg = (torch.rand(batch, 4, 16, 16, generator=g) * 60 + 1).to(dev)
gt_cls = (torch.rand(batch, 1, 16, 16, generator=g) > 0.8).float().to(dev)
gt_w = (torch.rand(batch, 16, 16, generator=g) > 0.9).float().to(dev)
net.step(tmpl, srch, gt_reg, gt_cls, gt_w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(steps):
    out = net.step(tmpl, srch, gt_reg, gt_cls, gt_w)
e1.record()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"mode={mode} two_streams={int(two)}: {1e3 * (time.perf_counter() - t0) / steps:.2f} ms/step wall, {e0.elapsed_time(e1) / steps:.2f} ms/step on the stream, "
      f"host issue {1e3 * t_issue / steps:.2f} ms/step, batch {batch}, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GB", file=sys.stderr)
# where the time goes: events at the phase boundaries of one more step (main stream)
net.phase_marks = []
net.step(tmpl, srch, gt_reg, gt_cls, gt_w)
torch.cuda.synchronize()
marks = net.phase_marks
for (n0, a), (n1, b) in zip(marks, marks[1:]):
    print(f"  {a.elapsed_time(b):7.2f} ms  {n1}", file=sys.stderr)
