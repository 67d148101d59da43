"""The optimiser side of a training step (reference: Adam(lr=1e-4), train/base_lightning_model.py:63-64): `fear_adam_step`
against torch.optim.Adam, `AdamHIP` on the network's kernel-layout parameters against torch.optim.Adam on the checker's
module, and the loop train -> state_dict -> export -> HIP inference."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(B=2, seed=9):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, 3, 128, 128, generator=g), torch.randn(B, 3, 256, 256, generator=g),
            torch.rand(B, 4, 16, 16, generator=g) * 60 + 1, (torch.rand(B, 1, 16, 16, generator=g) > 0.8).float(),
            (torch.rand(B, 16, 16, generator=g) > 0.85).float())


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_adam_operator_matches_torch(wd):
    import ctypes
    from feartracker_amd.train_head import _p, load_train_library
    lib = load_train_library()
    g = torch.Generator().manual_seed(3)
    n = 10007
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, weight_decay=wd)
    p = p0.clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (10.0 ** (step - 3))
        ref.grad = grad.clone()
        opt.step()
        gd = grad.cuda()
        assert lib.fear_adam_step(_p(p), _p(gd), _p(m), _p(v), n, 1e-3, 0.9, 0.999, 1e-8, wd, step, st) == 0
        torch.cuda.synchronize()
        err = float((p.cpu() - ref.detach()).abs().max())
        assert err <= 2e-7 * float(ref.detach().abs().max()), (step, err)
    assert lib.fear_adam_step(_p(p), _p(gd), _p(m), _p(v), n, 1e-3, 0.9, 0.999, 1e-8, wd, 0, st) == -2     # steps count from 1


def test_adam_on_kernel_layout_parameters_matches_torch_adam_on_the_module():
    """Same gradients (the checker's autograd) through AdamHIP and through torch.optim.Adam: every one of the 195 parameter
    tensors — depthwise taps stored [k*k][C], padded pointwise rows, the [16][28] stem — lands where torch puts it."""
    from feartracker_amd.optim import AdamHIP
    from feartracker_amd.train_net import FEARNetTrainHIP
    from oracle.fear_train_oracle import FEARNetTrainOracle, fear_loss, random_init_state
    sd = random_init_state(5)
    ora = FEARNetTrainOracle().train()
    ora.load_state_dict(sd, strict=False)
    net = FEARNetTrainHIP(sd, device=0)
    hip_opt = AdamHIP(net, lr=1e-3)
    ref_opt = torch.optim.Adam(ora.parameters(), lr=1e-3)
    tmpl, srch, gt_reg, gt_cls, gt_w = _batch()
    for it in range(2):
        ref_opt.zero_grad()
        bbox, cls = ora(tmpl, srch)
        lc, lr = fear_loss(bbox, cls, gt_reg, gt_cls, gt_w)
        (lc + lr).backward()
        grads = {n: p.grad.detach().clone() for n, p in ora.named_parameters()}
        ref_opt.step()
        hip_opt.step(grads)
        torch.cuda.synchronize()
        mine = net.state_dict()
        for n, p in ora.named_parameters():
            d = float((mine[n] - p.detach()).abs().max())
            assert mine[n].shape == p.shape and d <= 1e-6 * max(1.0, float(p.detach().abs().max())), (it, n, d)
    with pytest.raises(KeyError):
        hip_opt.step({k: v for k, v in grads.items() if k != "stem.conv.weight"})


def test_training_loop_reduces_the_loss_and_exports(tmp_path):
    """A few Adam steps of the HIP training step on one fixed batch bring the loss down; the trained state folds into a
    .fearw that the inference engine runs, matching the eval-mode forward of the same state."""
    from feartracker_amd import FEARNetHIP
    from feartracker_amd.export import export_training_state
    from feartracker_amd.optim import AdamHIP
    from feartracker_amd.train_net import FEARNetTrainHIP
    from oracle.fear_train_oracle import FEARNetTrainOracle, random_init_state
    net = FEARNetTrainHIP(random_init_state(5), device=0)
    opt = AdamHIP(net, lr=2e-3)
    tmpl, srch, gt_reg, gt_cls, gt_w = _batch(B=4, seed=21)
    losses = []
    for _ in range(8):
        out = net.step(tmpl, srch, gt_reg, gt_cls, gt_w)
        losses.append(float(out["loss_cls"]) + float(out["loss_reg"]))
        opt.step(out["grads"])
    assert all(l == l for l in losses) and losses[-1] < 0.97 * losses[0], losses          # measured: 1.680 -> 1.588
    assert all(b < a for a, b in zip(losses, losses[1:])), losses
    state = net.state_dict()
    path = os.path.join(tmp_path, "trained.fearw")
    export_training_state(state, path, payload="fp32")
    ora = FEARNetTrainOracle()
    ora.load_state_dict(state, strict=False)
    ora.eval()
    with torch.no_grad():
        bbox, cls = ora(tmpl, srch)
    eng = FEARNetHIP(path, device=0, max_batch=4)
    b, c = eng.track_maps(srch.cuda(), eng.get_features(tmpl.cuda()))
    eb = float((b.cpu() - bbox).abs().max() / bbox.abs().max())
    ec = float((c.cpu() - cls).abs().max() / cls.abs().max())
    assert eb < 1e-3 and ec < 1e-3, (eb, ec)
