import sys, time, torch
sys.path.insert(0, '/root/repo')
from feartracker_amd import FEARNetHIP, DEFAULT_WEIGHTS
from feartracker_amd.hip_backend import WEIGHTS_FEAR_M
def run(weights, B, tag):
    g = torch.Generator().manual_seed(3)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1) * 255.0
    inv = 1.0 / (torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1) * 255.0)
    x = ((torch.randint(0, 256, (B, 3, 256, 256), dtype=torch.uint8, generator=g).float() - mean) * inv).cuda()
    t = ((torch.randint(0, 256, (B, 3, 128, 128), dtype=torch.uint8, generator=g).float() - mean) * inv).cuda()
    outs = {}
    for on in (True, False):
        net = FEARNetHIP(weights, device=0, max_batch=B)
        net.set_small_pass(0)
        net.set_math(2)
        net.set_head_chain(on)
        z = net.get_features(t)
        names = [n for n, _, _ in net.plan(256, True)]
        b, c = net.track_maps(x, z)
        torch.cuda.synchronize()
        for _ in range(5): net.track_maps(x, z)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): net.track_maps(x, z)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
        outs[on] = (b.clone(), c.clone())
        print(tag, "head_chain", on, "ops", len(names), [n for n in names if "head" in n], f"{ms:.3f} ms  {B/ms:.1f} k crops/s")
    (b1, c1), (b0, c0) = outs[True], outs[False]
    print(tag, "rel diff bbox", float((b1 - b0).abs().max() / b0.abs().max()), "cls", float((c1 - c0).abs().max() / c0.abs().max()), "finite", bool(torch.isfinite(b1).all() and torch.isfinite(c1).all()))
run(DEFAULT_WEIGHTS, 256, "FEAR-XS bf16 B=256")
run(WEIGHTS_FEAR_M, 512, "FEAR-M bf16 B=512")
