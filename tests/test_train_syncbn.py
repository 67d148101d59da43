"""SyncBatchNorm operators (include/fear_train.h; the reference's multi-GPU backends train with sync_bn: True,
config/backend/{2,4}gpu.yaml -> trainer.py:52).  One GPU here, so the semantics are checked by playing two ranks on it: the
two halves of a batch reduce separately, their float64 sums are added (what the all-reduce does), and the result must equal
BatchNorm over the whole batch; the torch.distributed plumbing runs with a one-rank RCCL group."""
import ctypes
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_half_batches_with_added_sums_equal_full_batch_batchnorm():
    from feartracker_amd.train_head import _p, load_train_library
    lib = load_train_library()
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(5)
    M, C = 2 * 1536, 96
    x = (torch.randn(M, C, generator=g) * 2 + 0.7).to(dev)
    dy = torch.randn(M, C, generator=g).to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.rand(C, generator=g) - 0.5).to(dev)
    ws = torch.empty(lib.fear_train_workspace_bytes(M, C) // 4 + 1024, device=dev)
    wsb = ws.numel() * 4
    new = lambda *s: torch.empty(s, device=dev)
    # ---- whole batch, plain BatchNorm
    y, mean, rstd = new(M, C), new(C), new(C)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    assert lib.fear_bn_train_forward(_p(x), C, _p(gamma), _p(beta), _p(y), C, _p(mean), _p(rstd), _p(rm), _p(rv), 0.1, 1e-5, M, C, 1,
                                     _p(ws), wsb, st) == 0
    dx, dgamma, dbeta = new(M, C), new(C), new(C)
    assert lib.fear_bn_train_backward(_p(dy), C, _p(y), C, _p(x), C, _p(mean), _p(rstd), _p(gamma), _p(dx), C, _p(dgamma), _p(dbeta),
                                      M, C, _p(ws), wsb, st) == 0
    # ---- two "ranks": halves of the batch, sums added in between
    h = M // 2
    halves = [(x[:h].contiguous(), dy[:h].contiguous()), (x[h:].contiguous(), dy[h:].contiguous())]
    sums = [torch.empty(2 * C, dtype=torch.float64, device=dev) for _ in halves]
    for (xh, _), s in zip(halves, sums):
        assert lib.fear_bn_reduce(_p(xh), C, _p(s), h, C, _p(ws), wsb, st) == 0
    total = sums[0] + sums[1]
    outs = []
    for xh, _ in halves:
        yh, mh, rh = new(h, C), new(C), new(C)
        rmh, rvh = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        assert lib.fear_bn_forward_from_sums(_p(xh), C, _p(total), float(M), _p(gamma), _p(beta), _p(yh), C, _p(mh), _p(rh), _p(rmh),
                                             _p(rvh), 0.1, 1e-5, h, C, 1, st) == 0
        outs.append((yh, mh, rh, rmh, rvh))
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([o[0] for o in outs]), y)                  # same float64 sums up to the order of two additions:
    for o in outs:                                                           # the statistics agree to fp32 rounding, y bit for bit here
        for a, b in zip(o[1:], (mean, rstd, rm, rv)):
            assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())
    bsums = [torch.empty(2 * C, dtype=torch.float64, device=dev) for _ in halves]
    for (xh, dyh), o, s in zip(halves, outs, bsums):
        assert lib.fear_bn_backward_reduce(_p(dyh), C, _p(o[0]), C, _p(xh), C, _p(o[1]), _p(o[2]), _p(s), h, C, _p(ws), wsb, st) == 0
    btotal = bsums[0] + bsums[1]
    dxs, dgs, dbs = [], [], []
    for (xh, dyh), o, s in zip(halves, outs, bsums):
        dxh, dgh, dbh = new(h, C), new(C), new(C)
        assert lib.fear_bn_backward_from_sums(_p(dyh), C, _p(o[0]), C, _p(xh), C, _p(o[1]), _p(o[2]), _p(gamma), _p(btotal), float(M),
                                              _p(s), _p(dxh), C, _p(dgh), _p(dbh), _p(ws), wsb, h, C, st) == 0
        dxs.append(dxh); dgs.append(dgh); dbs.append(dbh)
    torch.cuda.synchronize()
    scale = float(dx.abs().max())
    assert float((torch.cat(dxs) - dx).abs().max()) <= 1e-5 * scale
    assert float((dgs[0] + dgs[1] - dgamma).abs().max()) <= 1e-5 * float(dgamma.abs().max())       # local sums add up to the
    assert float((dbs[0] + dbs[1] - dbeta).abs().max()) <= 1e-5 * float(dbeta.abs().max())         # full-batch parameter gradients
    # argument checks: count below the local row count is refused
    assert lib.fear_bn_forward_from_sums(_p(halves[0][0]), C, _p(total), float(h - 1), _p(gamma), _p(beta), _p(outs[0][0]), C,
                                         _p(outs[0][1]), _p(outs[0][2]), None, None, 0.1, 1e-5, h, C, 1, st) == -2


def test_sync_bn_step_with_a_one_rank_group_equals_the_plain_step():
    """The all-reduce plumbing (float64 sums on the device through RCCL, stream ordering against the operators): with one
    rank SyncBatchNorm is BatchNorm, so the whole training step must come out bit for bit."""
    import torch.distributed as dist
    from feartracker_amd.train_net import FEARNetTrainHIP, random_init_state
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sd = random_init_state(2)
        g = torch.Generator().manual_seed(4)
        B = 2
        tmpl, srch = torch.randn(B, 3, 128, 128, generator=g), torch.randn(B, 3, 256, 256, generator=g)
        gt_reg = torch.rand(B, 4, 16, 16, generator=g) * 60 + 1
        gt_cls = (torch.rand(B, 1, 16, 16, generator=g) > 0.8).float()
        gt_w = (torch.rand(B, 16, 16, generator=g) > 0.85).float()
        # (the layer-wise implementation of the trunk, its all-reduces between the passes; the block-fused default has its own test below)
        plain = FEARNetTrainHIP(sd, device=0, mode="layerwise").step(tmpl, srch, gt_reg, gt_cls, gt_w)
        net = FEARNetTrainHIP(sd, device=0, sync_bn=True, mode="layerwise")
        synced = net.step(tmpl, srch, gt_reg, gt_cls, gt_w)
        torch.cuda.synchronize()
        assert torch.equal(plain["bbox"], synced["bbox"]) and torch.equal(plain["cls"], synced["cls"])
        assert set(plain["grads"]) == set(synced["grads"])
        for k, v in plain["grads"].items():
            d = float((v - synced["grads"][k]).abs().max())
            assert d <= 1e-6 * max(float(v.abs().max()), 1e-12), (k, d)
        rs = net.running_stats()
        assert all(torch.isfinite(v).all() for v in rs.values())
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------------
# Block-fused operators (round 6): the BatchNorm reductions sit inside one C call, the ranks' all-reduce is the library's hook
# (include/fear_train.h fear_train_sync_bind, feartracker_amd/train_head.SyncHook).


class _TwoRanksOnOneGPU:
    """`.world` / `.all_reduce(tensor)` for two host threads that play two ranks on one device: both deposit their buffer, meet at
    a barrier, and each leaves with the sum — what an all-reduce over two ranks does, in the issue order of the two threads."""

    def __init__(self):
        import threading
        self.world = 2
        self.barrier = threading.Barrier(2)
        self.slots = {}
        self.calls = 0
        self.local = threading.local()

    def all_reduce(self, t: torch.Tensor) -> None:
        rank = self.local.rank
        torch.cuda.current_stream().synchronize()           # this rank's sums are complete
        self.slots[rank] = t
        self.barrier.wait()
        total = self.slots[0] + self.slots[1]               # (fixed order: both ranks get the same bits)
        torch.cuda.current_stream().synchronize()
        self.barrier.wait()                                  # both have read both buffers
        t.copy_(total)
        if rank == 0:
            self.calls += 1
        self.barrier.wait()


def test_block_mode_two_half_batches_through_the_hook_equal_the_full_batch():
    """The trunk's block-fused forward and backward (fear_stem_train_*, fear_irb_train_* incl. the virtual expansions whose
    statistics come from the input's Gram matrix, fear_pwbn_train_*) on two half batches — two host threads, each with its own
    network object and streams, exchanging sums through the hook — against the same operators on the whole batch: the same
    features to fp32 rounding (the float64 sums meet in another order), running statistics equal, 94 all-reduces, and the two
    ranks' parameter gradients add up to the full batch's — to 2e-2 only: the float64 sums of two halves round differently from
    the whole batch's in the last bit, and 50 BatchNorms of a random-init ReLU network at 4 crops amplify that on the way back
    (the deviation is 5e-6 from the neck down to the last-but-one block and jumps where a low-variance channel sits;
    tools/r6_syncdiag.py prints it per tensor).  What pins the hook's arithmetic is the block-level test below, against float64
    autograd on the whole batch at 1e-6."""
    import threading
    from feartracker_amd.train_net import FEARNetTrainHIP, random_init_state
    dev = torch.device("cuda:0")
    sd = random_init_state(3)
    g = torch.Generator().manual_seed(9)
    B = 4
    img = torch.randn(B, 3, 128, 128, generator=g).to(dev)
    dfeat = torch.randn(B * 64, 256, generator=g).to(dev)

    def run(net, x, dy, bound):
        with torch.cuda.device(dev):
            stream = torch.cuda.Stream(device=dev)
            stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(stream):
                ctxm = net.hook.bound(stream) if bound else _null()
                with ctxm:
                    feats, ctx = net._features_forward_b(x)
                    gbuf = torch.zeros(net._ptotal, dtype=torch.float32, device=dev)
                    net._features_backward_b(ctx, dy, gbuf)
                stream.synchronize()
        return feats, gbuf, net.running_stats()

    import contextlib
    _null = contextlib.nullcontext
    full = run(FEARNetTrainHIP(sd, device=0, mode="block"), img, dfeat, False)
    fake = _TwoRanksOnOneGPU()
    nets = [FEARNetTrainHIP(sd, device=0, mode="block", sync_bn=fake) for _ in range(2)]
    out, errs = [None, None], []

    def rank_main(r):
        try:
            fake.local.rank = r
            h = B // 2
            out[r] = run(nets[r], img[r * h:(r + 1) * h].contiguous(), dfeat[r * h * 64:(r + 1) * h * 64].contiguous(), True)
        except BaseException as exc:      # noqa: BLE001
            errs.append(exc)
            fake.barrier.abort()
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    assert fake.calls > 90          # stem + 16 blocks x (2 or 3) + neck BatchNorms, forward and backward
    feats = torch.cat([out[0][0], out[1][0]])
    scale = float(full[0].abs().max())
    assert float((feats - full[0]).abs().max()) <= 2e-5 * scale
    gsum = out[0][1] + out[1][1]
    gs = float(full[1].abs().max())
    assert float((gsum - full[1]).abs().max()) <= 2e-2 * gs, float((gsum - full[1]).abs().max()) / gs
    for r in range(2):              # both ranks track the statistics of ALL rows
        for k, v in full[2].items():
            if k.startswith("connect_model."):
                continue
            assert float((out[r][2][k] - v).abs().max()) <= 1e-5 * float(v.abs().max()) + 1e-6, k


def test_block_mode_sync_bn_step_with_a_one_rank_group_equals_the_plain_block_step():
    """FEARNetTrainHIP(mode="block", sync_bn=True) through a real RCCL group of one rank: every BatchNorm of the step goes local
    sums -> hook -> RCCL all-reduce -> finalize on its stream (two streams + the weight-gradient stream, as on one rank) and must
    come out bit for bit as the plain block step — with one rank the two-stage finalize adds one term."""
    import torch.distributed as dist
    from feartracker_amd.train_net import FEARNetTrainHIP, random_init_state
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sd = random_init_state(2)
        g = torch.Generator().manual_seed(4)
        B = 2
        tmpl, srch = torch.randn(B, 3, 128, 128, generator=g), torch.randn(B, 3, 256, 256, generator=g)
        gt_reg = torch.rand(B, 4, 16, 16, generator=g) * 60 + 1
        gt_cls = (torch.rand(B, 1, 16, 16, generator=g) > 0.8).float()
        gt_w = (torch.rand(B, 16, 16, generator=g) > 0.85).float()
        plain_net = FEARNetTrainHIP(sd, device=0, mode="block")
        plain = plain_net.step(tmpl, srch, gt_reg, gt_cls, gt_w)
        net = FEARNetTrainHIP(sd, device=0, mode="block", sync_bn=True)
        assert net.mode == "block" and net.hook is not None and net.two_streams
        synced = net.step(tmpl, srch, gt_reg, gt_cls, gt_w)
        torch.cuda.synchronize()
        assert torch.equal(plain["bbox"], synced["bbox"]) and torch.equal(plain["cls"], synced["cls"])
        assert set(plain["grads"]) == set(synced["grads"])
        for k, v in plain["grads"].items():
            assert torch.equal(v, synced["grads"][k]), k
        for k, v in plain_net.running_stats().items():
            assert torch.equal(v, net.running_stats()[k]), k
        # the binding is the step's: afterwards the streams run the one-rank form again
        again = FEARNetTrainHIP(sd, device=0, mode="block").step(tmpl, srch, gt_reg, gt_cls, gt_w)
        assert torch.equal(again["bbox"], plain["bbox"])
    finally:
        dist.destroy_process_group()


def test_sync_bind_argument_checks():
    from feartracker_amd.train_head import FearSync, _ALLREDUCE_FN, load_train_library
    lib = load_train_library()
    cb = _ALLREDUCE_FN(lambda *a: 0)
    buf = torch.zeros(2048, dtype=torch.float64, device="cuda:0")
    st = ctypes.c_void_p(torch.cuda.Stream().cuda_stream)
    assert lib.fear_train_sync_bind(st, ctypes.byref(FearSync(cb, None, None, 16384, 1))) == -1           # no buffer
    assert lib.fear_train_sync_bind(st, ctypes.byref(FearSync(cb, None, buf.data_ptr(), 1024, 1))) == -2  # buffer too small
    assert lib.fear_train_sync_bind(st, ctypes.byref(FearSync(cb, None, buf.data_ptr(), 16384, 0))) == -2  # world < 1
    assert lib.fear_train_sync_bind(st, ctypes.byref(FearSync(cb, None, buf.data_ptr(), 16384, 2))) == 0
    assert lib.fear_train_sync_bind(st, None) == 0
    assert lib.fear_train_sync_bind(st, None) == 0                                                           # unbinding twice is fine


SYNC_BLOCK_CASES = [
    # cin, cexp, cout, k, stride, expand, residual,   B (all ranks), H, flags
    ((16, 16, 16, 3, 1, 0, 1), 4, 32, 0),
    ((16, 96, 24, 3, 2, 1, 0), 4, 32, 4),        # virtual expansion: BatchNorm1's statistics from the all-reduced Gram matrix
    ((24, 144, 32, 5, 2, 1, 0), 2, 32, 4),
    ((32, 192, 32, 5, 1, 1, 1), 4, 16, 0),       # E-free BatchNorm1 backward (chosen by the call: 32 input channels)
    ((64, 384, 112, 5, 1, 1, 0), 6, 16, 0),
    ((112, 672, 112, 5, 1, 1, 1), 4, 8, 0),      # the template branch's last stage, 8 x 8 tiles
    ((112, 336, 112, 5, 1, 1, 1), 2, 16, 0),
]


@pytest.mark.parametrize("cfg,B,H,flags", SYNC_BLOCK_CASES,
                         ids=[f"{c[0]}x{c[1]}x{c[2]}k{c[3]}s{c[4]}_b{b}h{h}" + ("_virtual" if f else "") for c, b, h, f in SYNC_BLOCK_CASES])
def test_irb_block_on_two_ranks_through_the_hook_matches_full_batch_autograd(cfg, B, H, flags):
    """One inverted-residual block (fear_irb_train_forward / _backward) on the two halves of a batch — two host threads, each stream
    bound to the all-reduce hook — against torch autograd (float64) on the WHOLE batch: outputs and input gradients of both halves,
    the running statistics of both ranks, and the two ranks' parameter gradients added up (what DDP's gradient all-reduce does,
    up to the division by the world size)."""
    import threading
    from test_train_block import _rel, _torch_block
    from feartracker_amd.train_head import FearIrbBlock, FearIrbGrads, FearIrbSaved, SyncHook, _p, load_train_library
    lib = load_train_library()
    dev = torch.device("cuda:0")
    cin, cexp, cout, k, stride, expand, residual = cfg
    g = torch.Generator().manual_seed(300 + cin + cexp + k + stride + H)
    Ho = H // stride
    x = torch.randn(B, cin, H, H, generator=g, dtype=torch.float64, requires_grad=True)
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
    rows = lambda t: t.detach().permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
    fake = _TwoRanksOnOneGPU()
    res, errs_t = [None, None], []

    def rank_main(r):
        try:
            fake.local.rank = r
            h = B // 2
            sl = slice(r * h, (r + 1) * h)
            keep = []
            blk = FearIrbBlock()
            blk.cin, blk.cexp, blk.cout, blk.k, blk.stride, blk.expand, blk.residual, blk.flags = cin, cexp, cout, k, stride, expand, residual, flags
            w_pw = f32(p["w_pw"]) if expand else None
            w_dw, w_pwl = f32(p["w_dw"]), f32(p["w_pwl"])
            blk.w_pw, blk.w_dw, blk.w_pwl = (w_pw.data_ptr() if expand else None), w_dw.data_ptr(), w_pwl.data_ptr()
            gam, bet, rm, rv = [], [], [], []
            for i in range(3):
                gam.append(f32(p[f"g{i}"])); bet.append(f32(p[f"b{i}"]))
                rm.append(torch.zeros(chans[i], device=dev)); rv.append(torch.ones(chans[i], device=dev))
                blk.gamma[i], blk.beta[i], blk.running_mean[i], blk.running_var[i] = gam[i].data_ptr(), bet[i].data_ptr(), rm[i].data_ptr(), rv[i].data_ptr()
            ws = torch.empty(int(lib.fear_irb_workspace_bytes(ctypes.byref(blk), h, H, H)) // 4 + 64, device=dev)
            scratch = torch.empty(int(lib.fear_irb_scratch_floats(ctypes.byref(blk), h, H, H)) + 64, device=dev)
            sv = FearIrbSaved()
            e = torch.empty(h * H * H, cexp, device=dev) if expand and not flags & 4 else None
            d, pp = torch.empty(h * Ho * Ho, cexp, device=dev), torch.empty(h * Ho * Ho, cout, device=dev)
            vec = [torch.empty(4 * c, device=dev) for c in chans]
            sv.e, sv.d, sv.p = (e.data_ptr() if e is not None else None), d.data_ptr(), pp.data_ptr()
            for i in range(3):
                sv.vec[i] = vec[i].data_ptr()
            xd, dyd = f32(rows(x[sl])), f32(rows(dout[sl]))
            out = torch.empty(h * Ho * Ho, cout, device=dev)
            gr = FearIrbGrads()
            gw_pw = torch.full((cexp, cin), float("nan"), device=dev) if expand else None
            gw_dw, gw_pwl = torch.full((k * k, cexp), float("nan"), device=dev), torch.full((cout, cexp), float("nan"), device=dev)
            gr.w_pw, gr.w_dw, gr.w_pwl = (gw_pw.data_ptr() if expand else None), gw_dw.data_ptr(), gw_pwl.data_ptr()
            gg, gb = [], []
            for i in range(3):
                gg.append(torch.full((chans[i],), float("nan"), device=dev)); gb.append(torch.full((chans[i],), float("nan"), device=dev))
                gr.gamma[i], gr.beta[i] = gg[i].data_ptr(), gb[i].data_ptr()
            dx = torch.full((h * H * H, cin), float("nan"), device=dev)
            stream = torch.cuda.Stream(device=dev)
            hook = SyncHook(lib, fake, dev)
            with torch.cuda.stream(stream), hook.bound(stream):
                st = ctypes.c_void_p(stream.cuda_stream)
                assert lib.fear_irb_train_forward(ctypes.byref(blk), ctypes.byref(sv), _p(xd), _p(out), h, H, H, 0.1, 1e-5, _p(ws), ws.numel() * 4, st) == 0
                assert lib.fear_irb_train_backward(ctypes.byref(blk), ctypes.byref(sv), ctypes.byref(gr), _p(xd), _p(dyd), _p(dx), _p(scratch),
                                                   h, H, H, _p(ws), ws.numel() * 4, st, None) == 0
                stream.synchronize()
            assert hook.error is None, hook.error
            res[r] = dict(out=out, dx=dx, w_pw=gw_pw, w_dw=gw_dw, w_pwl=gw_pwl, gg=gg, gb=gb, rm=rm, rv=rv)
        except BaseException as exc:      # noqa: BLE001
            errs_t.append(exc)
            fake.barrier.abort()
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs_t, errs_t
    assert fake.calls == (6 if expand else 4)
    errs = {"out": _rel(torch.cat([res[0]["out"], res[1]["out"]]), rows(ref)),
            "dx": _rel(torch.cat([res[0]["dx"], res[1]["dx"]]), rows(x.grad)),
            "d w_dw": _rel(res[0]["w_dw"] + res[1]["w_dw"], p["w_dw"].grad),
            "d w_pwl": _rel(res[0]["w_pwl"] + res[1]["w_pwl"], p["w_pwl"].grad)}
    if expand:
        errs["d w_pw"] = _rel(res[0]["w_pw"] + res[1]["w_pw"], p["w_pw"].grad)
    for i in range(0 if expand else 1, 3):
        errs[f"d gamma{i}"] = _rel(res[0]["gg"][i] + res[1]["gg"][i], p[f"g{i}"].grad)
        errs[f"d beta{i}"] = _rel(res[0]["gb"][i] + res[1]["gb"][i], p[f"b{i}"].grad)
        for r in range(2):
            errs[f"rank {r} running_mean{i}"] = _rel(res[r]["rm"][i], stats[f"rm{i}"])
            errs[f"rank {r} running_var{i}"] = _rel(res[r]["rv"][i], stats[f"rv{i}"])
    print({k_: f"{v:.1e}" for k_, v in errs.items()})
    bad = {k_: v for k_, v in errs.items() if not v < 2e-4}
    assert not bad, bad
