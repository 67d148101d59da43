"""A/B of the two bf16-mode (FEAR_OPT_MATH = 2) plan options on one box: one-launch head (FEAR_OPT_HEAD_CHAIN) and bf16 activation
storage in the front of the trunk (FEAR_OPT_BF16_STORE).  usage (GPU box): python tools/hcb_ab.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feartracker_amd import FEARNetHIP, DEFAULT_WEIGHTS
from feartracker_amd.hip_backend import WEIGHTS_FEAR_M

def run(weights, B, tag):
    g = torch.Generator().manual_seed(3)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1) * 255.0
    inv = 1.0 / (torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1) * 255.0)
    x = ((torch.randint(0, 256, (B, 3, 256, 256), dtype=torch.uint8, generator=g).float() - mean) * inv).cuda()
    t = ((torch.randint(0, 256, (B, 3, 128, 128), dtype=torch.uint8, generator=g).float() - mean) * inv).cuda()
    ref = FEARNetHIP(weights, device=0, max_batch=B)
    ref.set_small_pass(0)
    z = ref.get_features(t)
    bf, cf = ref.track_maps(x, z)
    del ref
    for chain, store in ((False, False), (True, False), (True, True)):
        net = FEARNetHIP(weights, device=0, max_batch=B)
        net.set_small_pass(0)
        net.set_math(2)
        net.set_head_chain(chain)
        net.set_bf16_store(store)
        b, c = net.track_maps(x, z)
        torch.cuda.synchronize()
        for _ in range(5): net.track_maps(x, z)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): net.track_maps(x, z)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
        print(f"{tag}: head_chain {int(chain)} bf16_store {int(store)}: {len(net.plan(256, True))} launches, {ms:.3f} ms, {B / ms:.1f} k crops/s | vs fp32: bbox rel "
              f"{float((b - bf).abs().max() / bf.abs().max()):.3e}, cls abs {float((c - cf).abs().max()):.3e} (scale {float(cf.abs().max()):.1f}), finite "
              f"{bool(torch.isfinite(b).all() and torch.isfinite(c).all())}")
        del net

run(DEFAULT_WEIGHTS, 256, "FEAR-XS bf16 B=256")
run(WEIGHTS_FEAR_M, 512, "FEAR-M  bf16 B=512")
