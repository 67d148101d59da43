"""diagnostic: per-parameter deviation of two hooked half batches from the full batch (tests/test_train_syncbn.py)"""
import os, sys, threading, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_train_syncbn import _TwoRanksOnOneGPU
from feartracker_amd.train_net import FEARNetTrainHIP, random_init_state
dev = torch.device("cuda:0")
sd = random_init_state(3)
g = torch.Generator().manual_seed(9)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
img = torch.randn(B, 3, 128, 128, generator=g).to(dev)
dfeat = torch.randn(B * 64, 256, generator=g).to(dev)
def run(net, x, dy, bound):
    with torch.cuda.device(dev):
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream):
            with (net.hook.bound(stream) if bound else contextlib.nullcontext()):
                feats, ctx = net._features_forward_b(x)
                gbuf = torch.zeros(net._ptotal, dtype=torch.float32, device=dev)
                net._features_backward_b(ctx, dy, gbuf)
            stream.synchronize()
    return feats, gbuf, ctx
net_full = FEARNetTrainHIP(sd, device=0, mode="block")
full = run(net_full, img, dfeat, False)
full2 = run(FEARNetTrainHIP(sd, device=0, mode="block"), img, dfeat, False)
print("full vs full again: grad max diff", float((full[1] - full2[1]).abs().max()))
fake = _TwoRanksOnOneGPU()
nets = [FEARNetTrainHIP(sd, device=0, mode="block", sync_bn=fake) for _ in range(2)]
out = [None, None]
def rank_main(r):
    fake.local.rank = r
    h = B // 2
    out[r] = run(nets[r], img[r * h:(r + 1) * h].contiguous(), dfeat[r * h * 64:(r + 1) * h * 64].contiguous(), True)
ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
[t.start() for t in ths]; [t.join() for t in ths]
print("calls", fake.calls)
gsum = out[0][1] + out[1][1]
# forward statistics vectors per block
for bi, (blkf, blk0) in enumerate(zip(full[2][1]["blocks"], out[0][2][1]["blocks"])):
    vf, v0 = blkf[4][3], blk0[4][3]
    errs = []
    for a, b in zip(vf, v0):
        if a is None: errs.append(None); continue
        C = a.numel() // 4
        errs.append((float(((a[:C] - b[:C]).abs() / (a[:C].abs() + 1e-6)).max()), float(((a[C:2*C] - b[C:2*C]).abs() / a[C:2*C].abs()).max())))
    print("block", bi, "mean/rstd rel err per BN", errs)
for key, off in sorted(net_full._goff.items(), key=lambda kv: kv[1]):
    if key.startswith("connect_model."): continue
    nxt = min([o for o in net_full._goff.values() if o > off] + [net_full._ptotal])
    a, b = full[1][off:nxt], gsum[off:nxt]
    m = float(a.abs().max())
    print(f"{key:40s} max|g| {m:10.3e} rel dev {float((a - b).abs().max()) / max(m, 1e-30):.2e}")
