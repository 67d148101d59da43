"""development probe: the LDS-staged training GEMMs through the C ABI on one stream (fear_pwbn_train_forward / _backward), per shape
usage: FEAR_LIB=<library> python tools/gemm_ab.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from feartracker_amd.train_head import _p, load_train_library
lib = load_train_library()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
print("library:", os.environ.get("FEAR_LIB", "default"))
for (M, K, N) in [(32768, 112, 672), (32768, 672, 112), (32768, 64, 384), (32768, 384, 64), (32768, 256, 256), (131072, 32, 192), (131072, 192, 32), (8192, 112, 672), (8192, 672, 112)]:
    x = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    gam = (torch.rand(N, generator=g) + 0.5).to(dev); bet = torch.zeros(N, device=dev)
    raw, vec, out = torch.empty(M, N, device=dev), torch.empty(4 * N, device=dev), torch.empty(M, N, device=dev)
    dy = torch.randn(M, N, generator=g).to(dev)
    dw, dg, db, dx = torch.empty(N, K, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(M, K, device=dev)
    ws = torch.empty(int(lib.fear_pwbn_workspace_bytes(M, K, N)) // 4 + 64, device=dev)
    fwd = lambda: lib.fear_pwbn_train_forward(_p(x), K, _p(w), _p(gam), _p(bet), None, None, _p(raw), _p(vec), 1, _p(out), M, K, N, 0.1, 1e-5, _p(ws), ws.numel() * 4, None)
    bwd = lambda: lib.fear_pwbn_train_backward(_p(dy), _p(raw), _p(vec), 1, _p(x), K, _p(w), _p(gam), _p(dw), _p(dg), _p(db), _p(dx), M, K, N, _p(ws), ws.numel() * 4, None, None)
    res = []
    for fn in (fwd, bwd):
        for _ in range(5):
            assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 30 * 1e3)
    print(f"M {M:6d} K {K:3d} N {N:3d}: forward call {res[0]:7.1f} us   backward call {res[1]:7.1f} us   checksum {float(out.double().sum()):.6e} {float(dx.double().sum()):.6e}")
