"""Block-fused operators of the trunk's training step (include/fear_train.h `fear_irb_*`, `fear_pwbn_*`; csrc/fear_train_block.h)
against torch autograd on one inverted-residual block (model_training/model/blocks.py:22-35 over mobile_cv's conv-BN-ReLU units:
expand 1x1 + BN + ReLU, depthwise + BN + ReLU, project 1x1 + BN [+ input]) — every shape class of the FEAR-XS trunk (SURVEY.md
Appendix A): with / without expansion, 3x3 / 5x5, stride 1 / 2, residual or not, maps of 8 ... 64 pixels, ragged channel slabs.
The whole network in this mode is pinned by tests/test_train_head.py::test_whole_network_training_step_matches_autograd."""
import ctypes

import pytest
import torch
import torch.nn.functional as F


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _torch_block(x, p, cfg, train_stats):
    cin, cexp, cout, k, stride, expand, residual = cfg
    y = x
    units = []
    if expand:
        y = F.conv2d(y, p["w_pw"].view(cexp, cin, 1, 1))
        units.append(("pw", y))
        y = F.relu(F.batch_norm(y, train_stats["rm0"], train_stats["rv0"], p["g0"], p["b0"], True, 0.1, 1e-5))
    y = F.conv2d(y, p["w_dw"].t().reshape(cexp, 1, k, k), stride=stride, padding=k // 2, groups=cexp)
    y = F.relu(F.batch_norm(y, train_stats["rm1"], train_stats["rv1"], p["g1"], p["b1"], True, 0.1, 1e-5))
    y = F.conv2d(y, p["w_pwl"].view(cout, cexp, 1, 1))
    y = F.batch_norm(y, train_stats["rm2"], train_stats["rv2"], p["g2"], p["b2"], True, 0.1, 1e-5)
    return y + x if residual else y


CASES = [
    # cin, cexp, cout, k, stride, expand, residual,   B, H
    ((16, 16, 16, 3, 1, 0, 1), 2, 64),       # stage 1: no expansion, 16 channels = one ragged slab
    ((16, 96, 24, 3, 2, 1, 0), 3, 32),       # stage 2: the widest expansion of the large maps, stride 2
    ((24, 24, 24, 3, 1, 0, 1), 2, 32),       # e1 blocks: 24 channels = 1.5 slabs
    ((24, 144, 32, 5, 2, 1, 0), 2, 32),      # 5x5 stride 2, 144 = 9 slabs of 16
    ((32, 192, 32, 5, 1, 1, 1), 2, 16),      # 5x5 stride 1, residual, 32-channel slabs
    ((32, 192, 32, 3, 1, 1, 1), 3, 32),
    ((32, 192, 64, 5, 2, 1, 0), 2, 32),
    ((64, 384, 112, 5, 1, 1, 0), 3, 16),
    ((112, 672, 112, 5, 1, 1, 1), 2, 8),     # the template branch's last stage: the 16 x 16 tile overhangs the 8 x 8 map
    ((112, 336, 112, 5, 1, 1, 1), 5, 16),    # 336 = 21 slabs of 16, odd crop count
]


# ... and the expansions of the large maps once more in the E-free form of BatchNorm1's backward (FEAR_IRB_LINEAR_BN1: the call
# chooses it by itself from 10^5 rows up — the last case, whose 81 920 rows also take the first-generation GEMMs)
LIN_CASES = [(c, b, h, 1) for c, b, h in CASES if c[5]] + [((16, 96, 24, 3, 2, 1, 0), 5, 128, 1), ((32, 192, 32, 5, 1, 1, 1), 2, 16, 2)]
# ... and the 16 -> 96 expansion never written (FEAR_IRB_VIRTUAL_E): FearIrbSaved.e = NULL, BatchNorm1's statistics from the Gram matrix
LIN_CASES += [((16, 96, 24, 3, 2, 1, 0), 3, 32, 4), ((16, 96, 24, 3, 2, 1, 0), 5, 128, 4), ((16, 64, 16, 3, 2, 1, 0), 2, 24, 4),
              ((24, 144, 32, 5, 2, 1, 0), 2, 32, 4), ((32, 192, 64, 5, 2, 1, 0), 2, 32, 4), ((32, 128, 48, 3, 2, 1, 0), 3, 16, 4),
              ((20, 80, 24, 5, 2, 1, 0), 2, 24, 4)]
# ... and the narrow blocks' projection weight gradient summed inside the masked-gradient pass (FEAR_IRB_FUSE_W3)
LIN_CASES += [((16, 16, 16, 3, 1, 0, 1), 2, 64, 8), ((24, 24, 24, 3, 1, 0, 1), 2, 32, 8), ((24, 24, 24, 3, 1, 0, 1), 40, 64, 8)]
ALL_CASES = [(c, b, h, 0) for c, b, h in CASES] + LIN_CASES


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,B,H,flags", ALL_CASES,
                         ids=[f"{c[0]}x{c[1]}x{c[2]}k{c[3]}s{c[4]}_b{b}h{h}" + {0: "", 1: "_lin", 2: "_nolin", 4: "_virtual", 8: "_fusew3"}[f] for c, b, h, f in ALL_CASES])
def test_irb_block_forward_backward_vs_autograd(cfg, B, H, flags):
    _irb_case(cfg, B, H, flags, biased=False)


BIASED_CASES = [((16, 96, 24, 3, 2, 1, 0), 3, 32, 4), ((24, 144, 32, 5, 2, 1, 0), 2, 32, 4), ((32, 192, 64, 5, 2, 1, 0), 2, 32, 4),
                ((32, 192, 32, 5, 1, 1, 1), 2, 16, 0), ((16, 96, 24, 3, 2, 1, 0), 3, 32, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,B,H,flags", BIASED_CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}k{c[3]}s{c[4]}_flags{f}" for c, b, h, f in BIASED_CASES])
def test_irb_block_on_biased_correlated_inputs(cfg, B, H, flags):
    """a virtual expansion's BatchNorm1 variance is w^T G w / M - mean^2 with G = x^T x accumulated in fp32; the cases
    above feed zero-mean uncorrelated inputs, the benign case.  Real block inputs are post-ReLU / residual: here every channel is
    relu(noise) + 3 plus a component shared by all channels (mean ~ 4 x the spread, channels correlated at 0.5), on the virtual
    blocks, an E-free one and a plain one — the same 2e-4 against float64 autograd, running statistics included."""
    _irb_case(cfg, B, H, flags, biased=True)


def _irb_case(cfg, B, H, flags, biased):
    from feartracker_amd.train_head import FearIrbBlock, FearIrbGrads, FearIrbSaved, _p, load_train_library
    lib = load_train_library()
    dev = torch.device("cuda:0")
    cin, cexp, cout, k, stride, expand, residual = cfg
    g = torch.Generator().manual_seed(100 + cin + cexp + k + stride + H)
    Ho = H // stride
    x = torch.randn(B, cin, H, H, generator=g, dtype=torch.float64)
    if biased:
        shared = torch.randn(B, 1, H, H, generator=g, dtype=torch.float64)
        x = (torch.relu(x) * 0.7 + shared * 0.7 + 3.0)
    x.requires_grad_(True)
    p = {"w_dw": torch.randn(k * k, cexp, generator=g, dtype=torch.float64) * (2.0 / (k * k)) ** 0.5,
         "w_pwl": torch.randn(cout, cexp, generator=g, dtype=torch.float64) * (2.0 / cexp) ** 0.5}
    if expand:
        p["w_pw"] = torch.randn(cexp, cin, generator=g, dtype=torch.float64) * (2.0 / cin) ** 0.5
    chans = (cexp, cexp, cout)
    for i in range(3):
        p[f"g{i}"] = torch.rand(chans[i], generator=g, dtype=torch.float64) + 0.5
        p[f"b{i}"] = torch.randn(chans[i], generator=g, dtype=torch.float64) * 0.3
    for v in p.values():
        v.requires_grad_(True)
    stats = {}
    for i in range(3):
        stats[f"rm{i}"] = torch.zeros(chans[i], dtype=torch.float64)
        stats[f"rv{i}"] = torch.ones(chans[i], dtype=torch.float64)
    ref = _torch_block(x, p, cfg, stats)
    dout = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dout)

    keep = []

    def D(t):
        keep.append(t.detach().to(dev, torch.float32).contiguous())
        return keep[-1]

    rows = lambda t: t.detach().permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    blk = FearIrbBlock()
    blk.cin, blk.cexp, blk.cout, blk.k, blk.stride, blk.expand, blk.residual = cin, cexp, cout, k, stride, expand, residual
    blk.flags = flags
    w_pw = D(p["w_pw"]) if expand else None
    w_dw, w_pwl = D(p["w_dw"]), D(p["w_pwl"])
    blk.w_pw, blk.w_dw, blk.w_pwl = (w_pw.data_ptr() if expand else None), w_dw.data_ptr(), w_pwl.data_ptr()
    gam, bet, rm, rv = [], [], [], []
    for i in range(3):
        gam.append(D(p[f"g{i}"])); bet.append(D(p[f"b{i}"]))
        rm.append(torch.zeros(chans[i], device=dev)); rv.append(torch.ones(chans[i], device=dev))
        blk.gamma[i], blk.beta[i], blk.running_mean[i], blk.running_var[i] = gam[i].data_ptr(), bet[i].data_ptr(), rm[i].data_ptr(), rv[i].data_ptr()
    wsb = int(lib.fear_irb_workspace_bytes(ctypes.byref(blk), B, H, H))
    assert wsb > 0
    ws = torch.empty(wsb // 4 + 64, device=dev)
    scratch = torch.empty(int(lib.fear_irb_scratch_floats(ctypes.byref(blk), B, H, H)) + 64, device=dev)
    sv = FearIrbSaved()
    e = torch.empty(B * H * H, cexp, device=dev) if expand and not flags & 4 else None
    assert not flags & 4 or lib.fear_irb_virtual_ok(ctypes.byref(blk)) == 1
    d, pp = torch.empty(B * Ho * Ho, cexp, device=dev), torch.empty(B * Ho * Ho, cout, device=dev)
    vec = [torch.empty(4 * c, device=dev) for c in chans]
    sv.e, sv.d, sv.p = (e.data_ptr() if e is not None else None), d.data_ptr(), pp.data_ptr()
    for i in range(3):
        sv.vec[i] = vec[i].data_ptr()
    xd = D(rows(x))
    out = torch.empty(B * Ho * Ho, cout, device=dev)
    assert lib.fear_irb_train_forward(ctypes.byref(blk), ctypes.byref(sv), _p(xd), _p(out), B, H, H, 0.1, 1e-5, _p(ws), ws.numel() * 4, None) == 0
    torch.cuda.synchronize()
    errs = {"out": _rel(out, rows(ref))}
    for i in range(0 if expand else 1, 3):
        errs[f"running_mean{i}"] = _rel(rm[i], stats[f"rm{i}"])
        errs[f"running_var{i}"] = _rel(rv[i], stats[f"rv{i}"])
    gr = FearIrbGrads()
    gw_pw = torch.full((cexp, cin), float("nan"), device=dev) if expand else None
    gw_dw, gw_pwl = torch.full((k * k, cexp), float("nan"), device=dev), torch.full((cout, cexp), float("nan"), device=dev)
    gr.w_pw, gr.w_dw, gr.w_pwl = (gw_pw.data_ptr() if expand else None), gw_dw.data_ptr(), gw_pwl.data_ptr()
    gg, gb = [], []
    for i in range(3):
        gg.append(torch.full((chans[i],), float("nan"), device=dev)); gb.append(torch.full((chans[i],), float("nan"), device=dev))
        gr.gamma[i], gr.beta[i] = gg[i].data_ptr(), gb[i].data_ptr()
    dx = torch.full((B * H * H, cin), float("nan"), device=dev)
    assert lib.fear_irb_train_backward(ctypes.byref(blk), ctypes.byref(sv), ctypes.byref(gr), _p(xd), _p(D(rows(dout))), _p(dx), _p(scratch),
                                       B, H, H, _p(ws), ws.numel() * 4, None, None) == 0
    torch.cuda.synchronize()
    errs["dx"] = _rel(dx, rows(x.grad))
    errs["d w_dw"] = _rel(gw_dw, p["w_dw"].grad)
    errs["d w_pwl"] = _rel(gw_pwl, p["w_pwl"].grad)
    if expand:
        errs["d w_pw"] = _rel(gw_pw, p["w_pw"].grad)
    for i in range(0 if expand else 1, 3):
        errs[f"d gamma{i}"] = _rel(gg[i], p[f"g{i}"].grad)
        errs[f"d beta{i}"] = _rel(gb[i], p[f"b{i}"].grad)
    print({k_: f"{v:.1e}" for k_, v in errs.items()})
    bad = {k_: v for k_, v in errs.items() if not v < 2e-4}
    assert not bad, bad
    # the same call again: bit-identical (fixed-order reductions), and without the input gradient where the block allows it
    dx2 = torch.empty_like(dx)
    gw2 = torch.empty_like(gw_dw)
    gr.w_dw = gw2.data_ptr()
    # ... with the two pointwise weight gradients on a second stream (`wgrad_stream`): same numbers
    gw_pwl2 = torch.full_like(gw_pwl, float("nan"))
    gr.w_pwl = gw_pwl2.data_ptr()
    gw_pw2 = torch.full_like(gw_pw, float("nan")) if expand else None
    if expand:
        gr.w_pw = gw_pw2.data_ptr()
    aux = torch.cuda.Stream()
    assert lib.fear_irb_train_backward(ctypes.byref(blk), ctypes.byref(sv), ctypes.byref(gr), _p(xd), _p(keep[-1]), _p(dx2), _p(scratch),
                                       B, H, H, _p(ws), ws.numel() * 4, None, ctypes.c_void_p(aux.cuda_stream)) == 0
    torch.cuda.synchronize()
    assert torch.equal(dx2, dx) and torch.equal(gw2, gw_dw) and torch.equal(gw_pwl2, gw_pwl)
    assert not expand or torch.equal(gw_pw2, gw_pw)


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,relu,need_dx", [(2 * 64 * 64, 28, 16, 1, False), (3 * 16 * 16, 112, 256, 0, True), (1000, 24, 40, 1, True)])
def test_pwbn_unit_forward_backward_vs_autograd(M, K, N, relu, need_dx):
    """The stem (im2col rows, K = 28) and the AdjustLayer neck (blocks.py:75-88) as lone conv + BatchNorm [+ ReLU] units."""
    from feartracker_amd.train_head import _p, load_train_library
    lib = load_train_library()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5 + K + N)
    x = torch.randn(M, K, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(N, K, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    gamma = (torch.rand(N, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = (torch.randn(N, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    rm_ref, rv_ref = torch.zeros(N, dtype=torch.float64), torch.ones(N, dtype=torch.float64)
    y = F.batch_norm((x @ w.t()).t().reshape(1, N, M, 1), rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5).reshape(N, M).t()
    y = F.relu(y) if relu else y
    dy = torch.randn(M, N, generator=g, dtype=torch.float64)
    y.backward(dy)
    D = lambda t: t.detach().to(dev, torch.float32).contiguous()
    xd, wd, gd, bd, dyd = D(x), D(w), D(gamma), D(beta), D(dy)
    rm, rv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    raw, vec, out = torch.empty(M, N, device=dev), torch.empty(4 * N, device=dev), torch.empty(M, N, device=dev)
    ws = torch.empty(int(lib.fear_pwbn_workspace_bytes(M, K, N)) // 4 + 64, device=dev)
    assert lib.fear_pwbn_train_forward(_p(xd), K, _p(wd), _p(gd), _p(bd), _p(rm), _p(rv), _p(raw), _p(vec), relu, _p(out), M, K, N, 0.1, 1e-5,
                                       _p(ws), ws.numel() * 4, None) == 0
    dw, dg, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)
    dx = torch.empty(M, K, device=dev) if need_dx else None
    assert lib.fear_pwbn_train_backward(_p(dyd), _p(raw), _p(vec), relu, _p(xd), K, _p(wd), _p(gd), _p(dw), _p(dg), _p(db), _p(dx), M, K, N,
                                        _p(ws), ws.numel() * 4, None, None) == 0
    torch.cuda.synchronize()
    errs = {"out": _rel(out, y), "running_mean": _rel(rm, rm_ref), "running_var": _rel(rv, rv_ref), "dw": _rel(dw, w.grad),
            "dgamma": _rel(dg, gamma.grad), "dbeta": _rel(db, beta.grad)}
    if need_dx:
        errs["dx"] = _rel(dx, x.grad)
    print({k_: f"{v:.1e}" for k_, v in errs.items()})
    bad = {k_: v for k_, v in errs.items() if not v < 2e-4}
    assert not bad, bad


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,cin,cout,bias,ldx_pad,ldo_pad", [(3, 16, 256, 256, True, 0, 0), (2, 16, 320, 256, True, 0, 64), (2, 8, 64, 48, False, 16, 0),
                                                            (40, 16, 256, 256, True, 0, 0)])
def test_sepbn_layer_forward_backward_vs_autograd(B, H, cin, cout, bias, ldx_pad, ldo_pad):
    """The head's layer — SepConv (3x3 depthwise + pointwise, with or without biases) + BatchNorm + ReLU, model_training/model/
    blocks.py:97-101 / 115-119 / 151-161 — as one call per direction; input / output rows with a pitch (the [encode | correlation]
    concatenation); the weight gradients on a second stream are bit-identical to the in-line ones.  The last case's 10 240 rows reach
    the row-sliced weight-gradient kernels of the full batch."""
    from feartracker_amd.train_head import FearSepGrads, FearSepLayer, _p, load_train_library
    lib = load_train_library()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11 + cin + cout + B)
    M = B * H * H
    R = lambda *s, scale=1.0: (torch.randn(*s, generator=g, dtype=torch.float64) * scale)
    x = R(B, cin, H, H).requires_grad_(True)
    taps = R(9, cin, scale=0.4).requires_grad_(True)
    w = R(cout, cin, scale=(2.0 / cin) ** 0.5).requires_grad_(True)
    b_dw = R(cin, scale=0.3).requires_grad_(True) if bias else None
    b_pw = R(cout, scale=0.3).requires_grad_(True) if bias else None
    gamma = (torch.rand(cout, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = R(cout, scale=0.3).requires_grad_(True)
    rm_ref, rv_ref = torch.zeros(cout, dtype=torch.float64), torch.ones(cout, dtype=torch.float64)
    d_ref = F.conv2d(x, taps.t().reshape(cin, 1, 3, 3), b_dw, padding=1, groups=cin)
    y = F.relu(F.batch_norm(F.conv2d(d_ref, w.view(cout, cin, 1, 1), b_pw), rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5))
    dy = R(B, cout, H, H)
    y.backward(dy)
    rows = lambda t: t.detach().permute(0, 2, 3, 1).reshape(M, -1)
    D = lambda t: None if t is None else t.detach().to(dev, torch.float32).contiguous()
    ldx, ldo = cin + ldx_pad, cout + ldo_pad
    xd = torch.zeros(M, ldx, device=dev)
    xd[:, :cin] = D(rows(x))
    td, wd, bdw, bpw, gd, bd, dyd = D(taps), D(w), D(b_dw), D(b_pw), D(gamma), D(beta), D(rows(dy))
    rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    ptr = lambda t: None if t is None else t.data_ptr()
    L = FearSepLayer(cin, cout, ptr(td), ptr(bdw), ptr(wd), ptr(bpw), ptr(gd), ptr(bd), ptr(rm), ptr(rv))
    ws = torch.empty(int(lib.fear_sepbn_workspace_bytes(ctypes.byref(L), B, H, H)) // 4 + 64, device=dev)
    d, raw, vec = torch.empty(M, cin, device=dev), torch.empty(M, cout, device=dev), torch.empty(4 * cout, device=dev)
    out = torch.full((M, ldo), 7.0, device=dev)
    assert lib.fear_sepbn_train_forward(ctypes.byref(L), _p(xd), ldx, _p(d), _p(raw), _p(vec), _p(out), ldo, B, H, H, 0.1, 1e-5, _p(ws),
                                        ws.numel() * 4, None) == 0

    def backward(aux):
        gt, gw, gg, gb = torch.empty(9, cin, device=dev), torch.empty(cout, cin, device=dev), torch.empty(cout, device=dev), torch.empty(cout, device=dev)
        dd, coef, dx = torch.empty(M, cin, device=dev), torch.empty(4 * cout, device=dev), torch.empty(M, cin, device=dev)
        G = FearSepGrads(gt.data_ptr(), gw.data_ptr(), gg.data_ptr(), gb.data_ptr())
        assert lib.fear_sepbn_train_backward(ctypes.byref(L), ctypes.byref(G), _p(xd), ldx, _p(d), _p(raw), _p(vec), _p(dyd), _p(dd), _p(coef),
                                             _p(dx), B, H, H, _p(ws), ws.numel() * 4, None,
                                             ctypes.c_void_p(aux.cuda_stream) if aux is not None else None) == 0
        torch.cuda.synchronize()
        return gt, gw, gg, gb, dx
    gt, gw, gg, gb, dx = backward(None)
    errs = {"out": _rel(out[:, :cout], rows(y)), "running_mean": _rel(rm, rm_ref), "running_var": _rel(rv, rv_ref), "d": _rel(d, rows(d_ref)),
            "dtaps": _rel(gt, taps.grad), "dw": _rel(gw, w.grad), "dgamma": _rel(gg, gamma.grad), "dbeta": _rel(gb, beta.grad),
            "dx": _rel(dx, rows(x.grad))}
    print({k_: f"{v:.1e}" for k_, v in errs.items()})
    bad = {k_: v for k_, v in errs.items() if not v < 2e-4}
    assert not bad, bad
    assert ldo_pad == 0 or bool((out[:, cout:] == 7.0).all())        # nothing written beyond the layer's own columns
    if bias:                                                          # the biases' gradients vanish in front of the BatchNorm
        assert float(b_pw.grad.abs().max()) < 1e-9 and float(b_dw.grad.abs().max()) < 1e-9
    torch.cuda.synchronize()
    again = backward(torch.cuda.Stream(device=dev))
    for a, b in zip((gt, gw, gg, gb, dx), again):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("n,H", [(3, 64), (2, 256), (5, 30)])
def test_stem_on_the_image_forward_backward_vs_autograd(n, H):
    """The stem — 3x3 stride-2 conv 3 -> 16 + BatchNorm + ReLU — with its im2col rows gathered from the NCHW image by the GEMM and by the
    weight gradient (fear_stem_train_*): same numbers as the materialised rows (fear_stem_im2col + fear_pwbn_train_*), bit for bit
    in the forward, and autograd's gradients."""
    from feartracker_amd.train_head import _p, load_train_library
    lib = load_train_library()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21 + n + H)
    x = torch.randn(n, 3, H, H, generator=g, dtype=torch.float64)
    w = (torch.randn(16, 3, 3, 3, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    gamma = (torch.rand(16, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = (torch.randn(16, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    rm_ref, rv_ref = torch.zeros(16, dtype=torch.float64), torch.ones(16, dtype=torch.float64)
    y = F.relu(F.batch_norm(F.conv2d(x, w, stride=2, padding=1), rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5))
    Ho = H // 2
    M = n * Ho * Ho
    dy = torch.randn(n, 16, Ho, Ho, generator=g, dtype=torch.float64)
    y.backward(dy)
    rows = lambda t: t.detach().permute(0, 2, 3, 1).reshape(M, -1)
    D = lambda t: t.detach().to(dev, torch.float32).contiguous()
    xd, gd, bd, dyd = D(x), D(gamma), D(beta), D(rows(dy))
    w28 = torch.zeros(16, 28, device=dev)
    w28[:, :27] = D(w).reshape(16, 27)
    ws = torch.empty(int(lib.fear_stem_workspace_bytes(n, H, H)) // 4 + 64, device=dev)
    rm, rv = torch.zeros(16, device=dev), torch.ones(16, device=dev)
    raw, vec, out = torch.empty(M, 16, device=dev), torch.empty(64, device=dev), torch.empty(M, 16, device=dev)
    assert lib.fear_stem_train_forward(_p(xd), _p(w28), _p(gd), _p(bd), _p(rm), _p(rv), _p(raw), _p(vec), _p(out), n, H, H, 0.1, 1e-5,
                                       _p(ws), ws.numel() * 4, None) == 0
    # the materialised form
    col = torch.empty(M, 28, device=dev)
    assert lib.fear_stem_im2col(_p(xd), _p(col), n, H, H, None) == 0
    rm2, rv2 = torch.zeros(16, device=dev), torch.ones(16, device=dev)
    raw2, vec2, out2 = torch.empty(M, 16, device=dev), torch.empty(64, device=dev), torch.empty(M, 16, device=dev)
    assert lib.fear_pwbn_train_forward(_p(col), 28, _p(w28), _p(gd), _p(bd), _p(rm2), _p(rv2), _p(raw2), _p(vec2), 1, _p(out2), M, 28, 16,
                                       0.1, 1e-5, _p(ws), ws.numel() * 4, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(raw, raw2) and torch.equal(out, out2) and torch.equal(vec, vec2)
    dw, dg, db = torch.empty(16, 28, device=dev), torch.empty(16, device=dev), torch.empty(16, device=dev)
    assert lib.fear_stem_train_backward(_p(dyd), _p(raw), _p(vec), _p(xd), _p(gd), _p(dw), _p(dg), _p(db), n, H, H, _p(ws), ws.numel() * 4,
                                        None, None) == 0
    dw2, dg2, db2 = torch.empty(16, 28, device=dev), torch.empty(16, device=dev), torch.empty(16, device=dev)
    assert lib.fear_pwbn_train_backward(_p(dyd), _p(raw2), _p(vec2), 1, _p(col), 28, _p(w28), _p(gd), _p(dw2), _p(dg2), _p(db2), None, M, 28, 16,
                                        _p(ws), ws.numel() * 4, None, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2) and torch.equal(dg, dg2) and torch.equal(db, db2)
    errs = {"out": _rel(out, rows(y)), "running_mean": _rel(rm, rm_ref), "running_var": _rel(rv, rv_ref),
            "dw": _rel(dw[:, :27].reshape(16, 3, 3, 3), w.grad), "dgamma": _rel(dg, gamma.grad), "dbeta": _rel(db, beta.grad)}
    print({k_: f"{v:.1e}" for k_, v in errs.items()})
    bad = {k_: v for k_, v in errs.items() if not v < 2e-4}
    assert not bad, bad
    assert float(dw[:, 27].abs().max()) == 0.0
