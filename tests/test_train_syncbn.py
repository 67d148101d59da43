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
        # (SyncBatchNorm runs the layer-wise implementation of the trunk; the one-rank default is the block-fused one)
        plain = FEARNetTrainHIP(sd, device=0, mode="layerwise").step(tmpl, srch, gt_reg, gt_cls, gt_w)
        net = FEARNetTrainHIP(sd, device=0, sync_bn=True)
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
