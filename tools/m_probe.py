import os, sys, time
import torch
sys.path.insert(0, "/root/repo")
from feartracker_amd import FEARNetHIP
from oracle.fear_oracle import OracleNet
W = "/root/repo/feartracker_amd/weights/fear_m_synth.fearw"
net = FEARNetHIP(W, device=0, max_batch=256); net.set_small_pass(0)
ora = OracleNet(W)
g = torch.Generator().manual_seed(5)
def norm(u):
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1) * 255.0
    inv = 1.0 / (torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1) * 255.0)
    return (u.float()-mean)*inv
x = norm(torch.randint(0,256,(3,3,256,256),dtype=torch.uint8,generator=g)); t = norm(torch.randint(0,256,(3,3,128,128),dtype=torch.uint8,generator=g))
zr = ora.get_features(t); ref = ora.track(x, zr)
for mode in (0, 1, 2):
    net.set_math(mode)
    z = net.get_features(t.cuda()); b, c = net.track_maps(x.cuda(), z)
    eb = ((b.cpu()-ref["TARGET_REGRESSION_LABEL_KEY"]).abs()/ref["TARGET_REGRESSION_LABEL_KEY"].abs()).max().item()
    ec = (c.cpu()-ref["TARGET_CLASSIFICATION_KEY"]).abs().max().item()
    print("mode", mode, "bbox max elementwise rel", eb, "cls max abs", ec, "z rel", ((z.cpu()-zr).abs().max()/zr.abs().max()).item())
    print(len(net.plan(256, True)), "ops")
    B = 512 if mode == 2 else 256
    net.set_max_batch(B)
    xs = torch.randn(B,3,256,256, device="cuda"); zs = torch.randn(B,256,8,8, device="cuda")
    for _ in range(3): net.track_maps(xs, zs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): net.track_maps(xs, zs)
    torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/10
    fl = sum(f for _,f,_ in net.plan(256, True))
    print(f"mode {mode}: {B/dt:.0f} crops/s, {dt*1e3:.2f} ms/step, {fl/1e6:.1f} MFLOP/crop, {fl*B/dt/1e12:.1f} TF/s")
