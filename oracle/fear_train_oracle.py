"""CPU oracle of the FEAR TRAINING step (forward in train mode + FEARLoss + autograd).  TEST INFRASTRUCTURE ONLY.

Only `tests/` (and bench.py's reporting legs) may import this module; the product never does.

What is restated, and what pins it
----------------------------------
* Head + loss: `BoxTower` (model_training/model/blocks.py:129-194, with `SepConv` :45-72, `MatrixMobile` :91-105,
  `MobileCorrelation` :108-126), `AdjustLayer` (:75-88) and `FEARLoss` / `BoxLoss` / `calc_iou`
  (model_training/train/loss.py:13-96) as plain torch modules with the reference's parameter names.  PINNED by
  tests/golden/head_train_step.npz — outputs, losses and every gradient produced by the reference's own classes + torch
  autograd (tools/make_golden.py section 11); tests/test_train_head.py::test_training_oracle_head_matches_reference_fixture.
* Trunk: `FEARNet.feature_extractor` = fbnet_c stages[0:18] of the un-vendored `mobile_cv` dependency
  (facebookresearch/mobile-vision@51804a68, requirements.txt:8; model/blocks.py:22-35).  Its training-mode form — where the
  BatchNorms sit — is NOT recoverable from /root/reference (the shipped .mlmodel has them folded): this restatement puts a
  BatchNorm2d after every convolution (expand 1x1 + BN + ReLU, depthwise + BN + ReLU, project 1x1 + BN, the published
  FBNet-V2 inverted-residual block) on the block table of SURVEY.md Appendix A, bias-free convs.  **Trunk training parity is
  unpinned by the reference**; it is the oracle for the HIP trunk backward in the sense "torch autograd on the same graph".
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

# (cin, cexp, cout, k, stride, expand, residual): fbnet_c stages[1:18] as decoded from FEAR-XS-NoEmbs.mlmodel
TRUNK_BLOCKS: List[Tuple[int, int, int, int, int, bool, bool]] = [
    (16, 16, 16, 3, 1, False, True),
    (16, 96, 24, 3, 2, True, False),
    (24, 24, 24, 3, 1, False, True),
    (24, 24, 24, 3, 1, False, True),
    (24, 144, 32, 5, 2, True, False),
    (32, 96, 32, 5, 1, True, True),
    (32, 192, 32, 5, 1, True, True),
    (32, 192, 32, 3, 1, True, True),
    (32, 192, 64, 5, 2, True, False),
    (64, 192, 64, 5, 1, True, True),
    (64, 384, 64, 5, 1, True, True),
    (64, 384, 64, 5, 1, True, True),
    (64, 384, 112, 5, 1, True, False),
    (112, 672, 112, 5, 1, True, True),
    (112, 672, 112, 5, 1, True, True),
    (112, 336, 112, 5, 1, True, True),
]


class _ReluGivenMask(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x.clamp_min(0)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask.to(g.dtype), None


class MaskableReLU(nn.Module):
    """ReLU whose BACKWARD can be told which elements were active (`masks`: one bool tensor per upcoming call).

    ReLU's derivative is a tie-break at 0: two correct fp32 forwards that agree to 1e-6 still disagree on the sign of the odd
    pre-activation that sits within 1e-6 of zero (measured: 1 element in 262 144 of one layer), and that single element moves
    a BatchNorm bias gradient — a heavily cancelling sum — by percents.  The gradient-parity tests therefore evaluate the
    oracle's backward on the activity pattern of the implementation under test, after checking that the two forwards agree;
    with no masks queued this is a plain ReLU."""

    def __init__(self):
        super().__init__()
        self.masks = []

    def forward(self, x):
        if self.masks:
            return _ReluGivenMask.apply(x, self.masks.pop(0))
        return F.relu(x)


class ConvBN(nn.Module):
    def __init__(self, cin, cout, k, stride=1, groups=1, relu=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.act = MaskableReLU() if relu else None

    def forward(self, x):
        x = self.bn(self.conv(x))
        return self.act(x) if self.act is not None else x


class IRBlock(nn.Module):
    def __init__(self, cin, cexp, cout, k, stride, expand, residual):
        super().__init__()
        self.pw = ConvBN(cin, cexp, 1) if expand else None
        self.dw = ConvBN(cexp, cexp, k, stride=stride, groups=cexp)
        self.pwl = ConvBN(cexp, cout, 1, relu=False)
        self.residual = residual

    def forward(self, x):
        y = self.pw(x) if self.pw is not None else x
        y = self.pwl(self.dw(y))
        return x + y if self.residual else y


class SepConv(nn.Module):                      # blocks.py:45-72
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.depthwise = nn.Conv2d(cin, cin, 3, padding=1, groups=cin, bias=bias)
        self.pointwise = nn.Conv2d(cin, cout, 1, bias=bias)

    def forward(self, x):
        return self.pointwise(self.depthwise(x))


class BoxTowerOracle(nn.Module):               # blocks.py:129-194 with towernum=2, mobile=True
    def __init__(self, c=256, towernum=2):
        super().__init__()

        def seq(*mods):
            return nn.Sequential(*mods)

        self.cls_encode = nn.Module()
        self.cls_encode.matrix11_s = seq(SepConv(c, c, bias=False), nn.BatchNorm2d(c), MaskableReLU())
        self.reg_encode = nn.Module()
        self.reg_encode.matrix11_s = seq(SepConv(c, c, bias=False), nn.BatchNorm2d(c), MaskableReLU())
        self.cls_dw = nn.Module()
        self.cls_dw.enc = seq(SepConv(c + 64, c), nn.BatchNorm2d(c), MaskableReLU())
        self.reg_dw = nn.Module()
        self.reg_dw.enc = seq(SepConv(c + 64, c), nn.BatchNorm2d(c), MaskableReLU())
        tower, cls_tower = [], []
        for _ in range(towernum):
            tower += [SepConv(c, c), nn.BatchNorm2d(c), MaskableReLU()]
            cls_tower += [SepConv(c, c), nn.BatchNorm2d(c), MaskableReLU()]
        self.bbox_tower = seq(*tower)
        self.cls_tower = seq(*cls_tower)
        self.bbox_pred = SepConv(c, 4)
        self.cls_pred = SepConv(c, 1)
        self.adjust = nn.Parameter(0.1 * torch.ones(1))
        self.bias = nn.Parameter(torch.ones(1, 4, 1, 1))

    @staticmethod
    def _corr(z, x):                           # MobileCorrelation.forward, blocks.py:121-124
        b, c, w, h = x.size()
        s = torch.matmul(z.reshape(b, c, -1).permute(0, 2, 1), x.view(b, c, -1)).view(b, -1, w, h)
        return torch.cat([x, s], dim=1)

    def forward(self, search, kernel):
        cls_x = self.cls_encode.matrix11_s(search)
        reg_x = self.reg_encode.matrix11_s(search)
        cls_dw = self.cls_dw.enc(self._corr(kernel, cls_x))
        reg_dw = self.reg_dw.enc(self._corr(kernel, reg_x))
        x = torch.exp(self.adjust * self.bbox_pred(self.bbox_tower(reg_dw)) + self.bias)
        cls = 0.1 * self.cls_pred(self.cls_tower(cls_dw))
        return x, cls


def calc_iou(reg_target, pred, smooth=1.0):    # loss.py:13-23
    target_area = (reg_target[..., 0] + reg_target[..., 2]) * (reg_target[..., 1] + reg_target[..., 3])
    pred_area = (pred[..., 0] + pred[..., 2]) * (pred[..., 1] + pred[..., 3])
    w_i = torch.min(pred[..., 0], reg_target[..., 0]) + torch.min(pred[..., 2], reg_target[..., 2])
    h_i = torch.min(pred[..., 3], reg_target[..., 3]) + torch.min(pred[..., 1], reg_target[..., 1])
    inter = w_i * h_i
    return (inter + smooth) / (target_area + pred_area - inter + smooth)


def fear_loss(bbox, cls, gt_reg, gt_cls, gt_weight, coef_cls=1.0, coef_reg=1.0):
    """FEARLoss.forward (loss.py:45-96): returns (classification loss, regression loss)."""
    p = bbox.permute(0, 2, 3, 1).reshape(-1, 4)
    t = gt_reg.permute(0, 2, 3, 1).reshape(-1, 4)
    sel = torch.nonzero(gt_weight.reshape(-1) > 0).squeeze(1)
    reg = (1 - calc_iou(t[sel], p[sel])).mean()
    pred, label = cls.reshape(-1), gt_cls.reshape(-1)
    # loss.py:77-78 indexes with `.nonzero().squeeze()`: a selection of exactly one cell becomes a 0-dim index and
    # `_weighted_cls_loss` (loss.py:68-73) then returns the constant 0 for that half
    pos, neg = label.eq(1).nonzero().squeeze(), label.eq(0).nonzero().squeeze()
    bce = nn.BCEWithLogitsLoss()

    def half(sel):
        if sel.dim() == 0:
            return pred.new_zeros(())
        return bce(pred[sel], label[sel])

    lc = 0.5 * half(pos) + 0.5 * half(neg)
    return lc * coef_cls, reg * coef_reg


class FEARNetTrainOracle(nn.Module):
    """FEARNet.forward((template, search)) in training mode (fear_net.py:83-88): both crops through the shared trunk + neck
    (BatchNorm statistics per pass, template first), then the head."""

    def __init__(self):
        super().__init__()
        self.stem = ConvBN(3, 16, 3, stride=2)
        self.trunk = nn.ModuleList([IRBlock(*b) for b in TRUNK_BLOCKS])
        self.neck = nn.Module()
        self.neck.downsample = nn.Sequential(nn.Conv2d(112, 256, 1, bias=False), nn.BatchNorm2d(256))   # AdjustLayer, blocks.py:75-88
        self.connect_model = BoxTowerOracle()

    def get_features(self, x):
        x = self.stem(x)
        for b in self.trunk:
            x = b(x)
        return self.neck.downsample(x)

    def forward(self, template, search):
        z = self.get_features(template)
        x = self.get_features(search)
        return self.connect_model(x, z)


def random_init_state(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded random initialisation (BASELINE configs[4]: "random-init"): torch defaults for the convs, non-trivial BatchNorm
    affine parameters and running statistics so that every gradient path is exercised."""
    torch.manual_seed(seed)
    net = FEARNetTrainOracle()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.rand(m.bias.shape, generator=g) * 0.4 - 0.2)
                m.running_mean.copy_(torch.rand(m.bias.shape, generator=g) * 0.2 - 0.1)
                m.running_var.copy_(torch.rand(m.bias.shape, generator=g) * 0.5 + 0.75)
    return {k: v.detach().clone() for k, v in net.state_dict().items() if "num_batches_tracked" not in k}
