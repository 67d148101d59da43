"""Training step of the whole FEAR network on MI355X — BASELINE.json configs[4] "backbone + xcorr fwd/bwd, random-init".

`FEARNetTrainHIP.step(template, search, targets)` is `FEARNet.forward((template, search))` in training mode
(model_training/model/fear_net.py:83-88: both crops through the shared trunk + AdjustLayer neck, template first, BatchNorm
on the statistics of each pass; then `BoxTower`), `FEARLoss` (train/loss.py:45-96) and the backward pass to EVERY
parameter — what `FEARLightningModel._training_step` + `loss.backward()` compute (train/fear_lightning_model.py:60-66).
The head, the loss and their backward are `train_head.BoxTowerTrainHIP`; this module adds the trunk and the neck on the
same hand-written HIP operators (include/fear_train.h): 1x1 convolutions as MFMA GEMMs (forward / dgrad / wgrad), the
stem conv as im2col + the same GEMMs, depthwise 3x3 / 5x5 stride 1 / 2 forward, dgrad and wgrad, train-mode BatchNorm
forward / backward with fixed-order reductions.  Host code only sequences kernels and owns the parameters.  The default on
one rank (`mode="block"`) is one C-ABI call per inverted-residual block and direction (`fear_irb_train_*`; the stem on the
image: `fear_stem_train_*`; the neck: `fear_pwbn_train_*`) on three streams — DESIGN.md §7 N3 has the recipe and its numbers.

Trunk definition: the reference takes it from the un-vendored `mobile_cv` package (fbnet_c, model/blocks.py:22-35) and
ships only the BatchNorm-folded inference trace, so the TRAINING form of the trunk is restated — block table of SURVEY.md
Appendix A, every convolution bias-free and followed by a BatchNorm (expand 1x1 + BN + ReLU, depthwise + BN + ReLU,
project 1x1 + BN, residual where the block keeps shape) — with parameter names `stem.{conv,bn}`, `trunk.<i>.{pw,dw,pwl}.
{conv,bn}`, `neck.downsample.{0,1}`, `connect_model.*`.  Its gradient parity is pinned by torch autograd on the same graph
(the CPU checker under tests/), NOT by the reference; the head's is pinned by the reference itself.  Random initialisation is
the only use (configs[4] says so): trained `mobile_cv` checkpoints cannot be loaded without their key names.

Several ranks: every rank runs `step` on its share of the batch and `allreduce_gradients` averages the flat gradient buffer
with ONE RCCL all-reduce (≈1.37 M floats); `sync_bn=True` makes every BatchNorm a SyncBatchNorm over the group — the
reference's multi-GPU backends train that way (`sync_bn: True`, config/backend/{2,4}gpu.yaml -> trainer.py:52) — at the price
of one small float64 all-reduce per BatchNorm and direction; without it the statistics stay per rank.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import numpy as np
import torch

from .train_head import (BoxTowerTrainHIP, FearBnRunning, FearIrbBlock, FearIrbGrads, FearIrbSaved, GradDict, SyncBN, TrainError, _p,
                         load_train_library)

# (cin, cexp, cout, k, stride, expand, residual): fbnet_c stages[1:18]
TRUNK_BLOCKS = [
    (16, 16, 16, 3, 1, False, True), (16, 96, 24, 3, 2, True, False), (24, 24, 24, 3, 1, False, True),
    (24, 24, 24, 3, 1, False, True), (24, 144, 32, 5, 2, True, False), (32, 96, 32, 5, 1, True, True),
    (32, 192, 32, 5, 1, True, True), (32, 192, 32, 3, 1, True, True), (32, 192, 64, 5, 2, True, False),
    (64, 192, 64, 5, 1, True, True), (64, 384, 64, 5, 1, True, True), (64, 384, 64, 5, 1, True, True),
    (64, 384, 112, 5, 1, True, False), (112, 672, 112, 5, 1, True, True), (112, 672, 112, 5, 1, True, True),
    (112, 336, 112, 5, 1, True, True),
]


def random_init_state(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded random initialisation of every parameter of the training graph (BASELINE configs[4]: "random-init"):
    He-normal convolutions, BatchNorm weight 1 / bias 0 / running statistics 0 / 1, adjust = 0.1, bias = 1 like
    BoxTower.__init__ (blocks.py:170-172).  Keys as documented in the module docstring."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(key, cout, cin_g, k):
        sd[key] = torch.randn(cout, cin_g, k, k, generator=g) * (2.0 / (cin_g * k * k)) ** 0.5

    def bn(prefix, c):
        sd[prefix + ".weight"], sd[prefix + ".bias"] = torch.ones(c), torch.zeros(c)
        sd[prefix + ".running_mean"], sd[prefix + ".running_var"] = torch.zeros(c), torch.ones(c)

    conv("stem.conv.weight", 16, 3, 3)
    bn("stem.bn", 16)
    for i, (cin, cexp, cout, k, stride, expand, residual) in enumerate(TRUNK_BLOCKS):
        if expand:
            conv(f"trunk.{i}.pw.conv.weight", cexp, cin, 1)
            bn(f"trunk.{i}.pw.bn", cexp)
        conv(f"trunk.{i}.dw.conv.weight", cexp, 1, k)
        bn(f"trunk.{i}.dw.bn", cexp)
        conv(f"trunk.{i}.pwl.conv.weight", cout, cexp, 1)
        bn(f"trunk.{i}.pwl.bn", cout)
    conv("neck.downsample.0.weight", 256, 112, 1)
    bn("neck.downsample.1", 256)
    for enc, corr, tower, pred, pc in (("cls_encode.matrix11_s", "cls_dw.enc", "cls_tower", "cls_pred", 1),
                                       ("reg_encode.matrix11_s", "reg_dw.enc", "bbox_tower", "bbox_pred", 4)):
        for name, cin, cout, bias, bnp in ((enc + ".0", 256, 256, False, enc + ".1"), (corr + ".0", 320, 256, True, corr + ".1"),
                                           (tower + ".0", 256, 256, True, tower + ".1"), (tower + ".3", 256, 256, True, tower + ".4"),
                                           (pred, 256, pc, True, None)):
            h = "connect_model." + name
            conv(h + ".depthwise.weight", cin, 1, 3)
            conv(h + ".pointwise.weight", cout, cin, 1)
            if bias:
                sd[h + ".depthwise.bias"], sd[h + ".pointwise.bias"] = torch.zeros(cin), torch.zeros(cout)
            if bnp:
                bn("connect_model." + bnp, cout)
    sd["connect_model.adjust"], sd["connect_model.bias"] = 0.1 * torch.ones(1), torch.ones(1, 4, 1, 1)
    return sd


class _ConvBN:
    """conv (stem / pointwise / depthwise, no bias) + BatchNorm2d (train mode) [+ ReLU]: parameters in kernel layout."""

    def __init__(self, name: str, kind: str, sd: Dict[str, torch.Tensor], dev, k: int = 1, stride: int = 1, relu: bool = True,
                 conv_key: str = ".conv.weight", bn_key: str = ".bn"):
        self.name, self.kind, self.k, self.stride, self.relu = name, kind, k, stride, relu
        self.conv_key, self.bn_key = name + conv_key, name + bn_key
        w = sd[self.conv_key].float()
        self.cout = w.shape[0]
        if kind == "dw":
            self.cin = self.cout
            self.w = w.reshape(self.cout, k * k).t().contiguous().to(dev)                 # taps [k*k][C]
        elif kind == "stem":
            self.cin = 28
            w28 = torch.zeros(self.cout, 28)
            w28[:, :27] = w.reshape(self.cout, 27)
            self.w = w28.to(dev)
        else:
            self.cin = w.shape[1]
            self.w = w.reshape(self.cout, self.cin).contiguous().to(dev)
        self.gamma = sd[self.bn_key + ".weight"].float().to(dev)
        self.beta = sd[self.bn_key + ".bias"].float().to(dev)
        self.running_mean = sd[self.bn_key + ".running_mean"].float().clone().to(dev)
        self.running_var = sd[self.bn_key + ".running_var"].float().clone().to(dev)


FEAR_IRB_VIRTUAL_E = 4        # include/fear_train.h


class FEARNetTrainHIP:
    def __init__(self, state_dict: Dict[str, "np.ndarray | torch.Tensor"], device: int = 0, momentum: float = 0.1, eps: float = 1e-5,
                 coef_cls: float = 1.0, coef_reg: float = 1.0, sync_bn: bool = False, group=None, fused: Optional[bool] = None,
                 two_streams: bool = True, mode: Optional[str] = None, virtual_expansion: int = 32):
        if not torch.cuda.is_available():
            raise RuntimeError("FEARNetTrainHIP needs a ROCm GPU; there is no CPU fallback")
        self.lib = load_train_library()
        self.device = torch.device(f"cuda:{int(device)}")
        self.momentum, self.eps = momentum, eps
        # Three implementations of the trunk's conv + BatchNorm units, the same arithmetic (tests/test_train_head.py pins each
        # against autograd):
        #   "block"      (default on one rank) one C-ABI call per inverted-residual block and direction, csrc/fear_train_block.h:
        #                statistics in the producing pass, BatchNorm / its backward / ReLU masks formed on load, the two
        #                BatchNorms around the depthwise conv and its three gradients in one LDS-tiled pass
        #                With SyncBatchNorm (round 6) the ranks' all-reduces are made by the library's hook between a producer's
        #                float64 sums and their finalize (train_head.SyncHook, include/fear_train.h fear_train_sync_bind)
        #   "layerwise"  (fused=False) one operator per layer and direction; SyncBatchNorm: all-reduces between the reductions and the applies
        #   "fused"      (fused=True) round 3's per-unit fused operators, the memory-saving form of "layerwise"
        if mode is None:
            mode = "fused" if fused else ("layerwise" if fused is False else "block")
        if mode not in ("block", "layerwise", "fused"):
            raise ValueError(f"unknown mode {mode!r}")
        if mode == "fused" and sync_bn:
            raise ValueError("mode='fused' has no SyncBatchNorm form; use mode='block' (default) or 'layerwise' with sync_bn")
        self.mode = mode
        # block mode: expansions of up to this many input channels are never written where the library has the kernels for it
        # (FEAR_IRB_VIRTUAL_E, include/fear_train.h: the stride-2 blocks); 0 keeps every expansion saved
        self.virtual_expansion = 32 if virtual_expansion is True else int(virtual_expansion)
        fused = mode == "fused"
        # fused=True: the trunk runs on the fused conv + BatchNorm operators of include/fear_train.h — a BatchNorm'd activation
        # is never written, consumers apply it on load: 11 instead of 16 passes over every saved tensor and 13.7 instead of
        # 21.7 GB at 128 pairs, the same gradients (tests/test_train_head.py) — and, measured, NOT faster: 33.8 vs 31.6 ms of
        # kernel time per 128-pair step (profiles/r03_train_kernel_stats{,_fused}.csv).  The step's kernels are not bound by
        # the bytes the fusion removes: at 25 us per launch on average they are latency- and issue-bound, the activation
        # arithmetic added to the weight-gradient loads costs more (+1.9 ms) than the two removed passes save, and the
        # producers' per-workgroup partial sums add finalisation work.  So the default stays one kernel per layer and
        # direction; fused is the memory-saving mode (larger per-rank batches).
        self.fused = bool(fused)
        # two_streams: the BACKWARD of the template pass (a quarter of the search pass's work, the same launches) runs on a second
        # HIP stream next to the search pass's — the two are independent between the head and the final add of the shared
        # parameters' gradients, and their small kernels fill each other's tails.  Each lane has its own workspace.  SyncBatchNorm:
        # the layer-wise form keeps one stream; the block form keeps its streams — each is bound to a hook buffer of its own and
        # the host issues the collectives of all of them in one order, the same on every rank (train_head.SyncHook).
        self.two_streams = bool(two_streams) and (not sync_bn or mode == "block")
        self._side = None
        self._aux = None
        self._lane = 0
        self._ws_lanes = {}
        self.timing = None        # layerwise / fused modes: a list — every pointwise weight-gradient launch is bracketed with events and appended (no effect in block mode: its weight gradients are issued by the C side)
        sd = {k: torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v.detach().cpu() for k, v in state_dict.items()}
        dev = self.device
        self.stem = _ConvBN("stem", "stem", sd, dev, k=3, stride=2)
        self.blocks: List[dict] = []
        for i, (cin, cexp, cout, k, stride, expand, residual) in enumerate(TRUNK_BLOCKS):
            self.blocks.append(dict(
                pw=_ConvBN(f"trunk.{i}.pw", "pw", sd, dev) if expand else None,
                dw=_ConvBN(f"trunk.{i}.dw", "dw", sd, dev, k=k, stride=stride),
                pwl=_ConvBN(f"trunk.{i}.pwl", "pw", sd, dev, relu=False), residual=residual))
        self.neck = _ConvBN("neck.downsample", "pw", sd, dev, relu=False, conv_key=".0.weight", bn_key=".1")
        self.head = BoxTowerTrainHIP({k[len("connect_model."):]: v for k, v in sd.items() if k.startswith("connect_model.")},
                                     device=device, momentum=momentum, eps=eps, coef_cls=coef_cls, coef_reg=coef_reg,
                                     sync_bn=sync_bn, group=group, fused=self.mode != "layerwise")
        self.sync = self.head.sync                 # SyncBatchNorm over the data-parallel group (config/backend/*.yaml: sync_bn)
        self.hook = self.head.hook if self.mode == "block" else None      # block mode: the library's all-reduce hook (one per net: streams are bound once)
        self.last_contexts = None
        # the trunk runs twice per step (template, search): each pass writes its parameter gradients into one flat buffer
        # (kernel layouts, offsets below) and ONE add joins the two — not one add launch per parameter
        self._goff: Dict[str, int] = {}
        total = 0
        for L in self._trunk_layers():
            for key, n in ((L.conv_key, L.w.numel()), (L.bn_key + ".weight", L.cout), (L.bn_key + ".bias", L.cout)):
                self._goff[key] = total
                total += (n + 3) // 4 * 4                      # 16-byte aligned slots
        self._gtotal = total                                   # the trunk's share: what the two passes both write
        # every parameter (kernel layout) lives in ONE flat buffer — trunk first, then the head — and every gradient in a buffer
        # of the same layout: the optimiser is one launch over (param_flat, grad flat, moments), several ranks all-reduce the
        # gradient buffer as it stands
        for name, (t, _, _) in self.head.parameter_slots().items():
            self._goff["connect_model." + name] = total
            total += (t.numel() + 3) // 4 * 4
        self._ptotal = total
        self.param_flat = torch.zeros(total, dtype=torch.float32, device=dev)

        def home(name: str, t: torch.Tensor) -> torch.Tensor:
            v = self.param_flat[self._goff[name]: self._goff[name] + t.numel()].view(t.shape)
            v.copy_(t)
            return v
        for L in self._trunk_layers():
            L.w, L.gamma, L.beta = home(L.conv_key, L.w), home(L.bn_key + ".weight", L.gamma), home(L.bn_key + ".bias", L.beta)
        self.head.rehome_parameters(lambda name, t: home("connect_model." + name, t))
        self._gcur = None
        self.head._galloc = lambda name, *shape: self._gslot(self._gcur, "connect_model." + name, *shape)
        self._irb = None                                       # block mode: ctypes descriptors, built on first use
        self.phase_marks = None                                # set to [] to have `step` record events at its phase boundaries

    def _trunk_layers(self) -> List["_ConvBN"]:
        layers = [self.stem]
        for blk in self.blocks:
            layers += [L for L in (blk["pw"], blk["dw"], blk["pwl"]) if L is not None]
        return layers + [self.neck]

    # ------------------------------------------------------------------ plumbing
    def _check(self, st: int) -> None:
        if st != 0:
            self.head._check(st)

    def _stream(self):
        import ctypes
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _new(self, *shape) -> torch.Tensor:
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _workspace(self, rows: int):
        return self._lane_workspace(int(self.lib.fear_train_workspace_bytes(rows, 672)))

    def _lane_workspace(self, need: int):
        ws = self._ws_lanes.get(self._lane)
        if ws is None or ws.numel() * 4 < need:
            self._ws_lanes[self._lane] = None
            ws = self._ws_lanes[self._lane] = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.device)
        return _p(ws), ws.numel() * 4

    # ------------------------------------------------------------------ one conv + BN [+ ReLU]
    def _fwd(self, L: _ConvBN, x: torch.Tensor, B: int, H: int, saved: list, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: NHWC rows [B*H*H][cin] (stem: the im2col rows of the output grid).  Returns the activation rows (+ `residual`, the
        block input, when given: the skip connection is added by the BatchNorm's own apply kernel)."""
        lib, st = self.lib, self._stream()
        Ho = H // L.stride if L.kind == "dw" else H
        M = B * Ho * Ho
        ws, wsb = self._workspace(max(M, B * H * H))
        pre = self._new(M, L.cout)
        if L.kind == "dw":
            self._check(lib.fear_dw_forward(_p(x), L.cin, _p(L.w), None, _p(pre), L.cout, B, H, H, L.cin, L.k, L.stride, st))
        else:
            self._check(lib.fear_pw_forward(_p(x), L.cin, _p(L.w), None, _p(pre), L.cout, M, L.cin, L.cout, st))
        out, mean, rstd = self._new(M, L.cout), self._new(L.cout), self._new(L.cout)
        ab = None
        if self.sync is None:
            # the affine form a = gamma * rstd, b = beta - mean * a: y = fma(pre, a, b) — the backward recomputes the ReLU mask from
            # `pre` with the same fma and never reads `out` again (2 of the 7 passes over a ReLU layer's tensors on the way back)
            ab = (self._new(L.cout), self._new(L.cout))
            self._check(lib.fear_bn_train_forward_ab(_p(pre), L.cout, _p(L.gamma), _p(L.beta), 1 if L.relu else 0, _p(residual), L.cout,
                                                     _p(out), L.cout, _p(mean), _p(rstd), _p(ab[0]), _p(ab[1]), _p(L.running_mean),
                                                     _p(L.running_var), self.momentum, self.eps, M, L.cout, ws, wsb, st))
        else:
            # SyncBatchNorm: the same arithmetic with the float64 sums added over the ranks between the two halves
            sums = torch.empty(2 * L.cout, dtype=torch.float64, device=self.device)
            self._check(lib.fear_bn_reduce(_p(pre), L.cout, _p(sums), M, L.cout, ws, wsb, st))
            self.sync.all_reduce(sums)
            ab = (self._new(L.cout), self._new(L.cout))
            self._check(lib.fear_bn_finalize(_p(sums), float(M) * self.sync.world, _p(L.gamma), _p(L.beta), _p(mean), _p(rstd), _p(ab[0]),
                                             _p(ab[1]), _p(L.running_mean), _p(L.running_var), self.momentum, self.eps, L.cout, st))
            self._check(lib.fear_bn_act(_p(pre), L.cout, _p(ab[0]), _p(ab[1]), 1 if L.relu else 0, _p(residual), L.cout, _p(out), L.cout,
                                        M, L.cout, st))
        saved.append((L, x, pre, out, mean, rstd, B, H, ab))
        return out

    def _gslot(self, gbuf: torch.Tensor, key: str, *shape) -> torch.Tensor:
        n = int(np.prod(shape))
        off = self._goff[key]
        return gbuf[off: off + n].view(*shape)

    def _bwd(self, rec, dy: torch.Tensor, gbuf: torch.Tensor, need_dx: bool = True,
             add: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """Backward of one conv + BN [+ ReLU]; the parameter gradients go to their slots of `gbuf` (kernel layouts).  `add`: a
        tensor added to the input gradient (the skip connection's gradient) — inside the dgrad GEMM for pointwise units."""
        L, x, pre, out, mean, rstd, B, H, ab = rec
        lib, st = self.lib, self._stream()
        Ho = H // L.stride if L.kind == "dw" else H
        M = B * Ho * Ho
        ws, wsb = self._workspace(max(M, B * H * H))
        dpre = self._new(M, L.cout)
        dgamma, dbeta = self._gslot(gbuf, L.bn_key + ".weight", L.cout), self._gslot(gbuf, L.bn_key + ".bias", L.cout)
        if self.sync is None:
            self._check(lib.fear_bn_train_backward_x(_p(dy), L.cout, _p(pre), L.cout, _p(ab[0]), _p(ab[1]), 1 if L.relu else 0, _p(mean),
                                                     _p(rstd), _p(L.gamma), _p(dpre), L.cout, _p(dgamma), _p(dbeta), M, L.cout, ws, wsb, st))
        else:
            sums = torch.empty(2 * L.cout, dtype=torch.float64, device=self.device)
            self._check(lib.fear_bn_backward_reduce_x(_p(dy), L.cout, _p(pre), L.cout, _p(ab[0]), _p(ab[1]), 1 if L.relu else 0, _p(mean),
                                                      _p(rstd), _p(sums), M, L.cout, ws, wsb, st))
            local = sums.clone()
            self.sync.all_reduce(sums)
            self._check(lib.fear_bn_backward_apply_x(_p(dy), L.cout, _p(pre), L.cout, _p(ab[0]), _p(ab[1]), 1 if L.relu else 0, _p(mean),
                                                     _p(rstd), _p(L.gamma), _p(sums), float(M) * self.sync.world, _p(local), _p(dpre), L.cout,
                                                     _p(dgamma), _p(dbeta), ws, wsb, M, L.cout, st))
        dx = None
        if L.kind == "dw":
            dtaps = self._gslot(gbuf, L.conv_key, L.k * L.k, L.cout)
            self._check(lib.fear_dw_backward_weight(_p(dpre), L.cout, _p(x), L.cin, _p(dtaps), ws, wsb, B, H, H, L.cin, L.k, L.stride, st))
            if need_dx:
                dx = self._new(B * H * H, L.cin)
                self._check(lib.fear_dw_backward_data(_p(dpre), L.cout, _p(L.w), _p(dx), L.cin, B, H, H, L.cin, L.k, L.stride, st))
                if add is not None:
                    self._check(lib.fear_add(_p(dx), _p(add), _p(dx), dx.numel(), st))
        else:
            dw = self._gslot(gbuf, L.conv_key, L.cout, L.cin)
            if self.timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self._check(lib.fear_pw_backward_weight(_p(dpre), L.cout, _p(x), L.cin, _p(dw), ws, wsb, M, L.cin, L.cout, st))
            if self.timing is not None:
                e1.record()
                # algorithmic bytes of dW[n][k] = sum_m dY[m][n] X[m][k]: both operands read once, dW written once
                self.timing.append((e0, e1, 4.0 * (M * L.cout + M * L.cin + L.cout * L.cin), 2.0 * M * L.cout * L.cin))
            if L.kind != "stem":
                if need_dx:
                    dx = self._new(M, L.cin)
                    self._check(lib.fear_pw_backward_data(_p(dpre), L.cout, _p(L.w), _p(add), L.cin if add is not None else 0, _p(dx),
                                                          L.cin, M, L.cin, L.cout, st))
        return dx

    def _add(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        out = self._new(*a.shape)
        self._check(self.lib.fear_add(_p(a), _p(b), _p(out), a.numel(), self._stream()))
        return out

    # ------------------------------------------------------------------ fused conv + BN units
    def _workspace_stats(self, rows: int, channels: int):
        return self._lane_workspace(max(int(self.lib.fear_train_stats_workspace_bytes(rows, channels)),
                                        int(self.lib.fear_train_workspace_bytes(rows, 672))))

    def _fwd_f(self, L: _ConvBN, x: torch.Tensor, x_act, B: int, H: int, saved: list):
        """One conv + BatchNorm unit on the fused operators.  x: rows [B*H*H][cin] — a previous unit's RAW conv output when
        `x_act` = (a, b, relu) names the activation to apply on load, else a plain tensor.  Writes this unit's raw output and
        returns (raw, (a, b, relu)): whoever consumes it applies the BatchNorm [+ ReLU] itself."""
        lib, st = self.lib, self._stream()
        Ho = H // L.stride if L.kind == "dw" else H
        M = B * Ho * Ho
        ws, wsb = self._workspace_stats(max(M, B * H * H), L.cout)
        pre = self._new(M, L.cout)
        sums = torch.empty(2 * L.cout, dtype=torch.float64, device=self.device)
        ia, ib, irelu = (_p(x_act[0]), _p(x_act[1]), int(x_act[2])) if x_act is not None else (None, None, 0)
        if L.kind == "dw":
            self._check(lib.fear_dw_forward_stats(_p(x), L.cin, ia, ib, irelu, _p(L.w), _p(pre), L.cout, B, H, H, L.cin, L.k, L.stride,
                                                  _p(sums), ws, wsb, st))
        else:
            self._check(lib.fear_pw_forward_stats(_p(x), L.cin, ia, ib, irelu, _p(L.w), _p(pre), L.cout, M, L.cin, L.cout, _p(sums),
                                                  ws, wsb, st))
        count = float(M)
        if self.sync is not None:
            self.sync.all_reduce(sums)
            count *= self.sync.world
        mean, rstd, a, b = self._new(L.cout), self._new(L.cout), self._new(L.cout), self._new(L.cout)
        self._check(lib.fear_bn_finalize(_p(sums), count, _p(L.gamma), _p(L.beta), _p(mean), _p(rstd), _p(a), _p(b), _p(L.running_mean),
                                         _p(L.running_var), self.momentum, self.eps, L.cout, st))
        act = (a, b, 1 if L.relu else 0)
        saved.append(dict(L=L, x=x, x_act=x_act, pre=pre, mean=mean, rstd=rstd, act=act, B=B, H=H))
        return pre, act

    def _materialise(self, pre: torch.Tensor, act, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = self._new(*pre.shape)
        C = pre.shape[1]
        self._check(self.lib.fear_bn_act(_p(pre), C, _p(act[0]), _p(act[1]), int(act[2]), _p(residual), C, _p(out), C, pre.shape[0], C,
                                         self._stream()))
        return out

    def _bwd_f(self, rec: dict, dy: torch.Tensor, gbuf: torch.Tensor, need_dx: bool = True,
               add: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """Backward of one fused unit given dy = gradient of its ACTIVATION; parameter gradients go to their slots of `gbuf`.
        `add` (pointwise units): a tensor added to the input gradient in the same kernel (the residual branch's gradient)."""
        L, x, x_act, pre, mean, rstd, act, B, H = (rec[k] for k in ("L", "x", "x_act", "pre", "mean", "rstd", "act", "B", "H"))
        lib, st = self.lib, self._stream()
        Ho = H // L.stride if L.kind == "dw" else H
        M = B * Ho * Ho
        ws, wsb = self._workspace_stats(max(M, B * H * H), max(L.cout, L.cin))
        sums = torch.empty(2 * L.cout, dtype=torch.float64, device=self.device)
        a, b, relu = act
        self._check(lib.fear_bn_backward_reduce_x(_p(dy), L.cout, _p(pre), L.cout, _p(a), _p(b), relu, _p(mean), _p(rstd), _p(sums), M,
                                                  L.cout, ws, wsb, st))
        local, count = sums, float(M)
        if self.sync is not None:
            local = sums.clone()
            self.sync.all_reduce(sums)
            count *= self.sync.world
        dpre = self._new(M, L.cout)
        dgamma, dbeta = self._gslot(gbuf, L.bn_key + ".weight", L.cout), self._gslot(gbuf, L.bn_key + ".bias", L.cout)
        self._check(lib.fear_bn_backward_apply_x(_p(dy), L.cout, _p(pre), L.cout, _p(a), _p(b), relu, _p(mean), _p(rstd), _p(L.gamma),
                                                 _p(sums), count, _p(local), _p(dpre), L.cout, _p(dgamma), _p(dbeta), ws, wsb, M, L.cout, st))
        ia, ib, irelu = (_p(x_act[0]), _p(x_act[1]), int(x_act[2])) if x_act is not None else (None, None, 0)
        dx = None
        if L.kind == "dw":
            dtaps = self._gslot(gbuf, L.conv_key, L.k * L.k, L.cout)
            self._check(lib.fear_dw_backward_weight_act(_p(dpre), L.cout, _p(x), L.cin, ia, ib, irelu, _p(dtaps), ws, wsb, B, H, H, L.cin,
                                                        L.k, L.stride, st))
            if need_dx:
                dx = self._new(B * H * H, L.cin)
                self._check(lib.fear_dw_backward_data(_p(dpre), L.cout, _p(L.w), _p(dx), L.cin, B, H, H, L.cin, L.k, L.stride, st))
                if add is not None:
                    self._check(lib.fear_add(_p(dx), _p(add), _p(dx), dx.numel(), st))
        else:
            dw = self._gslot(gbuf, L.conv_key, L.cout, L.cin)
            self._check(lib.fear_pw_backward_weight_act(_p(dpre), L.cout, _p(x), L.cin, ia, ib, irelu, _p(dw), ws, wsb, M, L.cin, L.cout, st))
            if L.kind != "stem" and need_dx:
                dx = self._new(M, L.cin)
                self._check(lib.fear_pw_backward_data(_p(dpre), L.cout, _p(L.w), _p(add), L.cin if add is not None else 0, _p(dx), L.cin,
                                                      M, L.cin, L.cout, st))
        return dx

    def _features_forward_f(self, img: torch.Tensor):
        """Fused form of `_features_forward`: img (B,3,H,H) NCHW -> (feature rows [B*(H/16)^2][256], context for the backward)."""
        B, H = img.shape[0], img.shape[2]
        saved: list = []
        h = H // 2
        col = self._new(B * h * h, 28)
        self._check(self.lib.fear_stem_im2col(_p(img), _p(col), B, H, H, self._stream()))
        pre, act = self._fwd_f(self.stem, col, None, B, h, saved)
        x = self._materialise(pre, act)                              # block 1 adds it back as its residual: a real tensor
        block_recs = []
        for blk in self.blocks:
            start = len(saved)
            y, yact = x, None
            if blk["pw"] is not None:
                y, yact = self._fwd_f(blk["pw"], y, yact, B, h, saved)
            y, yact = self._fwd_f(blk["dw"], y, yact, B, h, saved)
            h = h // blk["dw"].stride
            y, yact = self._fwd_f(blk["pwl"], y, yact, B, h, saved)
            x = self._materialise(y, yact, x if blk["residual"] else None)     # BN of the projection (+ residual): the block output
            block_recs.append((start, len(saved), blk["residual"]))
        pre, act = self._fwd_f(self.neck, x, None, B, h, saved)
        return self._materialise(pre, act), (saved, block_recs, B, h)

    def _features_backward_f(self, ctx, dfeat: torch.Tensor, gbuf: torch.Tensor) -> None:
        saved, block_recs, B, h = ctx
        d = self._bwd_f(saved[-1], dfeat, gbuf)                      # neck
        for start, end, residual in reversed(block_recs):
            dres = d if residual else None
            for i in range(end - 1, start - 1, -1):
                d = self._bwd_f(saved[i], d, gbuf, add=dres if i == start else None)     # the block's first unit also takes the skip's gradient
        self._bwd_f(saved[0], d, gbuf, need_dx=False)                # stem: the image needs no gradient

    # ------------------------------------------------------------------ block-fused trunk (mode "block")
    def _irb_descriptors(self):
        """ctypes mirrors of FearIrbBlock for the 16 blocks (the parameter tensors are views of param_flat: stable addresses)."""
        if self._irb is None:
            descs = []
            for (cin, cexp, cout, k, stride, expand, residual), blk in zip(TRUNK_BLOCKS, self.blocks):
                d = FearIrbBlock()
                d.cin, d.cexp, d.cout, d.k, d.stride, d.expand, d.residual = cin, cexp, cout, k, stride, int(expand), int(residual)
                units = (blk["pw"], blk["dw"], blk["pwl"])
                d.w_pw = units[0].w.data_ptr() if expand else None
                d.w_dw, d.w_pwl = units[1].w.data_ptr(), units[2].w.data_ptr()
                for i, L in enumerate(units):
                    if L is None:
                        continue
                    d.gamma[i], d.beta[i] = L.gamma.data_ptr(), L.beta.data_ptr()
                    d.running_mean[i], d.running_var[i] = L.running_mean.data_ptr(), L.running_var.data_ptr()
                if cin <= self.virtual_expansion and self.lib.fear_irb_virtual_ok(ctypes.byref(d)):
                    d.flags = FEAR_IRB_VIRTUAL_E          # the 16 -> 96 expansion of the 128 x 128 map: never written (include/fear_train.h)
                descs.append(d)
            self._irb = descs
        return self._irb

    def _block_buffers(self, B: int, H: int):
        """(workspace pointer, bytes, scratch tensor) of this lane for a pass over B crops of H x H pixels."""
        import ctypes
        lib = self.lib
        need, scratch = int(lib.fear_pwbn_workspace_bytes(B * (H // 2) ** 2, 28, 16)), 0
        h = H // 2
        for i, d in enumerate(self._irb_descriptors()):
            nb, ns = int(lib.fear_irb_workspace_bytes(ctypes.byref(d), B, h, h)), int(lib.fear_irb_scratch_floats(ctypes.byref(d), B, h, h))
            if nb == 0 or ns == 0:
                # (the block operators index their tensors with 32-bit offsets: B * H * W * cexp * 4 bytes < 2^31, i.e. up to
                #  341 pairs per rank on the 96-channel 128 x 128 map)
                raise TrainError(f"trunk block {i} ({d.cin}->{d.cexp}->{d.cout} at {h}x{h}) does not support {B} crops per pass in mode='block'; "
                                 "use a smaller per-rank batch or mode='layerwise'")
            need, scratch = max(need, nb), max(scratch, ns)
            h //= d.stride
        need = max(need, int(lib.fear_pwbn_workspace_bytes(B * h * h, 112, 256)))
        ws, wsb = self._lane_workspace(need)
        key = ("scratch", self._lane)
        sc = self._ws_lanes.get(key)
        if sc is None or sc.numel() < scratch:
            self._ws_lanes[key] = None
            sc = self._ws_lanes[key] = torch.empty(scratch, dtype=torch.float32, device=self.device)
        return ws, wsb, sc

    def _features_forward_b(self, img: torch.Tensor, defer_running: bool = False):
        """Block-fused form of `_features_forward`: one call per block (csrc/fear_train_block.h).  `defer_running`: leave the
        BatchNorm running statistics alone and list (vec, rows, unit) in the context instead — `_apply_running` updates them
        later (the search pass, while the template pass runs on the other stream)."""
        import ctypes
        lib, st = self.lib, self._stream()
        B, H = img.shape[0], img.shape[2]
        ws, wsb, _ = self._block_buffers(B, H)
        h = H // 2
        S = self.stem
        run = (lambda L: (None, None)) if defer_running else (lambda L: (_p(L.running_mean), _p(L.running_var)))
        pending = []
        stem_raw, stem_vec, x = self._new(B * h * h, 16), self._new(4 * 16), self._new(B * h * h, 16)
        # (the stem's im2col rows are gathered from the image by the GEMM and, in the backward, by the weight gradient)
        self._check(lib.fear_stem_train_forward(_p(img), _p(S.w), _p(S.gamma), _p(S.beta), *run(S), _p(stem_raw), _p(stem_vec), _p(x),
                                                B, H, H, self.momentum, self.eps, ws, wsb, st))
        pending.append((stem_vec, B * h * h, S))
        recs = [dict(L=S, pre=stem_raw, act=(stem_vec[32:48], stem_vec[48:64], 1), B=B, H=h)]     # what relu_patterns reads
        blocks = []
        descs = self._irb_descriptors()
        if defer_running:
            if getattr(self, "_irb_norun", None) is None:
                self._irb_norun = []
                for d in descs:
                    c = FearIrbBlock.from_buffer_copy(d)
                    for i in range(3):
                        c.running_mean[i], c.running_var[i] = None, None
                    self._irb_norun.append(c)
            descs = self._irb_norun
        for d, blk in zip(descs, self.blocks):
            ho = h // d.stride
            sv = FearIrbSaved()
            virt = bool(d.flags & FEAR_IRB_VIRTUAL_E)
            e = self._new(B * h * h, d.cexp) if d.expand and not virt else None
            dd, pp = self._new(B * ho * ho, d.cexp), self._new(B * ho * ho, d.cout)
            vec = [self._new(4 * d.cexp) if d.expand else None, self._new(4 * d.cexp), self._new(4 * d.cout)]
            sv.e, sv.d, sv.p = (e.data_ptr() if e is not None else None), dd.data_ptr(), pp.data_ptr()
            for i in range(3):
                sv.vec[i] = vec[i].data_ptr() if vec[i] is not None else None
            out = self._new(B * ho * ho, d.cout)
            self._check(lib.fear_irb_train_forward(ctypes.byref(d), ctypes.byref(sv), _p(x), _p(out), B, h, h, self.momentum, self.eps, ws, wsb, st))
            C = d.cexp
            if d.expand:
                # (a virtual expansion has no saved pre-activation: relu_patterns forms x W1^T itself)
                recs.append(dict(L=blk["pw"], pre=e if not virt else (x, blk["pw"].w), act=(vec[0][2 * C: 3 * C], vec[0][3 * C:], 1), B=B, H=h))
                pending.append((vec[0], B * h * h, blk["pw"]))
            recs.append(dict(L=blk["dw"], pre=dd, act=(vec[1][2 * C: 3 * C], vec[1][3 * C:], 1), B=B, H=h))
            pending += [(vec[1], B * ho * ho, blk["dw"]), (vec[2], B * ho * ho, blk["pwl"])]
            blocks.append((d, sv, x, h, (e, dd, pp, vec)))          # (the tensors are kept alive next to their pointers)
            x, h = out, ho
        N = self.neck
        neck_raw, neck_vec, feats = self._new(B * h * h, 256), self._new(4 * 256), self._new(B * h * h, 256)
        self._check(lib.fear_pwbn_train_forward(_p(x), 112, _p(N.w), _p(N.gamma), _p(N.beta), *run(N), _p(neck_raw),
                                                _p(neck_vec), 0, _p(feats), B * h * h, 112, 256, self.momentum, self.eps, ws, wsb, st))
        pending.append((neck_vec, B * h * h, N))
        return feats, (recs, dict(B=B, H=H, img=img, stem=(stem_raw, stem_vec), blocks=blocks, neck=(x, neck_raw, neck_vec, h),
                                  pending=pending if defer_running else []))

    def _apply_running(self, ctx) -> None:
        """The deferred running-statistics updates of a `_features_forward_b(..., defer_running=True)` pass, on the current stream."""
        st = self._stream()
        world = self.sync.world if self.sync is not None else 1      # (SyncBatchNorm: the statistics are those of all ranks' rows)
        pending = ctx[1]["pending"]
        if not pending:
            return
        items = (FearBnRunning * len(pending))()
        for it, (vec, rows, L) in zip(items, pending):
            vec.record_stream(torch.cuda.current_stream(self.device))
            it.vec, it.running_mean, it.running_var, it.C, it.count = vec.data_ptr(), L.running_mean.data_ptr(), L.running_var.data_ptr(), L.cout, float(rows) * world
        self._check(self.lib.fear_bn_running_update_multi(items, len(pending), self.momentum, self.eps, st))      # one launch for all of them

    def _features_backward_b(self, ctx, dfeat: torch.Tensor, gbuf: torch.Tensor, aux=None) -> None:
        """`aux`: a torch stream for the pointwise weight gradients (they do not feed the chain of input gradients): every block
        then gets a scratch of its own, and the caller joins `aux` before the gradients are used."""
        import ctypes
        lib, st = self.lib, self._stream()
        c = ctx[1]
        B, H = c["B"], c["H"]
        ws, wsb, scratch = self._block_buffers(B, H)
        aux_p = ctypes.c_void_p(aux.cuda_stream) if aux is not None else None
        if aux is not None:
            # one scratch per block (the gradient tensors a block leaves there are still being read by its weight gradients on
            # `aux` while the next blocks run): carved out of one allocation of the summed sizes
            sizes, h = [], H // 2
            for desc in self._irb_descriptors():
                sizes.append((int(lib.fear_irb_scratch_floats(ctypes.byref(desc), B, h, h)) + 63) // 64 * 64)
                h //= desc.stride
            key = ("scratch_all", self._lane)
            big = self._ws_lanes.get(key)
            if big is None or big.numel() < sum(sizes):
                self._ws_lanes[key] = None
                big = self._ws_lanes[key] = torch.empty(sum(sizes), dtype=torch.float32, device=self.device)
            offs = np.cumsum([0] + sizes)
            per_block = [big[offs[i]: offs[i + 1]] for i in range(len(sizes))]
            big.record_stream(aux)
            gbuf.record_stream(aux)
        g = lambda key, n: _p(self._gslot(gbuf, key, n))
        N = self.neck
        x_neck, neck_raw, neck_vec, h = c["neck"]
        d = self._new(B * h * h, 112)
        # (the neck's weight gradient stays in line: a lone unit's coefficient vectors live in the shared workspace, so a call with a
        #  weight-gradient stream first waits for everything queued there — the head's weight gradients at this point)
        self._check(lib.fear_pwbn_train_backward(_p(dfeat), _p(neck_raw), _p(neck_vec), 0, _p(x_neck), 112, _p(N.w), _p(N.gamma),
                                                 g(N.conv_key, 256 * 112), g(N.bn_key + ".weight", 256), g(N.bn_key + ".bias", 256), _p(d),
                                                 B * h * h, 112, 256, ws, wsb, st, None))
        for bi, ((desc, sv, x, hin, _keep), blk) in enumerate(zip(reversed(c["blocks"]), reversed(self.blocks))):
            gr = FearIrbGrads()
            units = (blk["pw"], blk["dw"], blk["pwl"])
            for i, L in enumerate(units):
                if L is None:
                    continue
                w = self._gslot(gbuf, L.conv_key, L.w.numel()).data_ptr()
                if i == 0:
                    gr.w_pw = w
                elif i == 1:
                    gr.w_dw = w
                else:
                    gr.w_pwl = w
                gr.gamma[i] = self._gslot(gbuf, L.bn_key + ".weight", L.cout).data_ptr()
                gr.beta[i] = self._gslot(gbuf, L.bn_key + ".bias", L.cout).data_ptr()
            dx = self._new(B * hin * hin, desc.cin)
            sc = per_block[len(self.blocks) - 1 - bi] if aux is not None else scratch
            if aux is not None:
                d.record_stream(aux)           # (read by this block's weight gradient on `aux` after this loop has dropped it)
            self._check(lib.fear_irb_train_backward(ctypes.byref(desc), ctypes.byref(sv), ctypes.byref(gr), _p(x), _p(d), _p(dx), _p(sc),
                                                    B, hin, hin, ws, wsb, st, aux_p))
            d = dx
        S = self.stem
        stem_raw, stem_vec = c["stem"]
        hs = H // 2
        # the stem's weight gradient in line as well, on a workspace of its own: the last blocks' weight gradients are still running on
        # the weight-gradient stream (on the shared workspace's row-slice region) when the chain of input gradients ends here — the
        # pass's last kernel runs beside them instead of behind them
        key = ("stem", self._lane)
        need = int(lib.fear_stem_workspace_bytes(B, H, H))
        sw = self._ws_lanes.get(key)
        if sw is None or sw.numel() * 4 < need:
            self._ws_lanes[key] = None
            sw = self._ws_lanes[key] = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.device)
        self._check(lib.fear_stem_train_backward(_p(d), _p(stem_raw), _p(stem_vec), _p(c["img"]), _p(S.gamma), g(S.conv_key, 16 * 28),
                                                 g(S.bn_key + ".weight", 16), g(S.bn_key + ".bias", 16), B, H, H, _p(sw), sw.numel() * 4, st, None))

    # ------------------------------------------------------------------ trunk + neck
    def _features_forward(self, img: torch.Tensor):
        """img (B,3,H,H) NCHW -> (feature rows [B*(H/16)^2][256], saved records for the backward)."""
        B, H = img.shape[0], img.shape[2]
        saved: list = []
        h = H // 2
        col = self._new(B * h * h, 28)
        self._check(self.lib.fear_stem_im2col(_p(img), _p(col), B, H, H, self._stream()))
        x = self._fwd(self.stem, col, B, h, saved)
        block_recs = []
        for blk in self.blocks:
            start = len(saved)
            y = x
            if blk["pw"] is not None:
                y = self._fwd(blk["pw"], y, B, h, saved)
            y = self._fwd(blk["dw"], y, B, h, saved)
            h = h // blk["dw"].stride
            y = self._fwd(blk["pwl"], y, B, h, saved, residual=x if blk["residual"] else None)
            block_recs.append((start, len(saved), blk["residual"]))
            x = y
        feats = self._fwd(self.neck, x, B, h, saved)
        return feats, (saved, block_recs, B, h)

    def _features_backward(self, ctx, dfeat: torch.Tensor, gbuf: torch.Tensor) -> None:
        saved, block_recs, B, h = ctx
        d = self._bwd(saved[-1], dfeat, gbuf)                        # neck
        for start, end, residual in reversed(block_recs):
            dres = d if residual else None
            for i in range(end - 1, start - 1, -1):
                d = self._bwd(saved[i], d, gbuf, add=dres if i == start else None)     # the block's first unit also takes the skip's gradient
        self._bwd(saved[0], d, gbuf, need_dx=False)                  # stem: the image needs no gradient

    # ------------------------------------------------------------------ the step
    @torch.no_grad()
    def step(self, template: torch.Tensor, search: torch.Tensor, gt_reg: torch.Tensor, gt_cls: torch.Tensor,
             gt_weight: torch.Tensor) -> Dict[str, object]:
        """template (B,3,128,128), search (B,3,256,256) normalised fp32 NCHW; targets as `BoxTowerTrainHIP.step`.
        Returns {"loss_cls", "loss_reg", "bbox", "cls", "grads": {parameter name: gradient}}."""
        dev = self.device
        t = template.to(dev, torch.float32).contiguous()
        s = search.to(dev, torch.float32).contiguous()
        B = s.shape[0]
        if tuple(t.shape) != (B, 3, 128, 128) or tuple(s.shape) != (B, 3, 256, 256):
            raise ValueError("expected template (B,3,128,128) and search (B,3,256,256)")
        with torch.cuda.device(dev):
            if self.two_streams and self._side is None:
                self._side = torch.cuda.Stream(device=dev)
                self.head.side_stream = self._side         # the head runs its two towers on the two streams as well
            # SyncBatchNorm in block mode: the streams that carry BatchNorms are bound to the library's all-reduce hook for the step
            # (the third stream has weight gradients only)
            import contextlib
            bound = self.hook.bound(torch.cuda.current_stream(dev), self._side if self.two_streams else None) if self.hook is not None else contextlib.nullcontext()
            with bound:
                out, grads = self._step_on_device(t, s, B, gt_reg, gt_cls, gt_weight)
        return {"loss_cls": out["loss_cls"], "loss_reg": out["loss_reg"], "bbox": out["bbox"], "cls": out["cls"], "grads": grads}

    def _step_on_device(self, t, s, B, gt_reg, gt_cls, gt_weight):
        dev = self.device
        if True:
            st = self._stream()
            ffwd = {"block": self._features_forward_b, "fused": self._features_forward_f, "layerwise": self._features_forward}[self.mode]
            fbwd = {"block": self._features_backward_b, "fused": self._features_backward_f, "layerwise": self._features_backward}[self.mode]
            # every gradient of the step lives in one buffer in param_flat's layout (the template pass's trunk gradients behind it)
            gall = torch.zeros(self._ptotal + self._gtotal, dtype=torch.float32, device=dev)      # fresh per step: the caller keeps `grads`
            gflat = (gall[: self._ptotal], gall[self._ptotal:])
            self._gcur = gflat[0]
            main = torch.cuda.current_stream(dev)
            side = self._side if self.two_streams else None
            marks = [] if self.phase_marks is not None else None      # (tools/train_prof.py: where the step's time goes)

            def mark(name):
                if marks is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(main)
                    marks.append((name, e))
            mark("start")
            if side is not None and self.mode == "block":
                # both trunk passes at once: the template pass (a quarter of the work, launch-bound small maps) on the side stream
                # under the search pass's bandwidth-bound kernels.  Both update the shared trunk's BatchNorm running statistics,
                # template first (torch's two forward calls): the search pass leaves them alone and its updates follow the template
                # pass on the side stream, next to the head
                side.wait_stream(main)
                t.record_stream(side)
                with torch.cuda.stream(side):
                    self._lane = 1
                    try:
                        zrows, zctx = ffwd(t)
                    finally:
                        self._lane = 0
                    z_done = side.record_event()
                xrows, xctx = ffwd(s, defer_running=True)
                x_done = main.record_event()
                with torch.cuda.stream(side):
                    side.wait_event(x_done)
                    self._apply_running(xctx)
                main.wait_event(z_done)
                zrows.record_stream(main)
            else:
                # (the FORWARD passes stay in order on one stream: both update the shared trunk's BatchNorm running statistics,
                # template first — torch's two forward calls — and that read-modify-write must not race)
                zrows, zctx = ffwd(t)                                    # template first, like FEARNet.forward
                xrows, xctx = ffwd(s)
            mark("trunk forward (search pass; template pass beside it)")
            z = self._new(B, 256, 8, 8)
            self._check(self.lib.fear_nhwc_to_nchw(_p(zrows), _p(z), B, 256, 64, 256, 0, st))
            # the head's weight gradients share the third stream with the trunk's (joined once, after the trunk's backward)
            use_aux = side is not None and self.mode == "block"
            if use_aux and self._aux is None:
                self._aux = torch.cuda.Stream(device=dev)
            self.head.aux_stream, self.head.aux_join = (self._aux, False) if use_aux else (None, True)
            if use_aux:
                self._aux.wait_stream(main)                         # (the gradient buffer's zeros)
                gall.record_stream(self._aux)
            out = self.head.step_rows(xrows, z, gt_reg, gt_cls, gt_weight)      # (the search features stay pixel rows both ways)
            grads = GradDict({"connect_model." + k: v for k, v in out["grads"].items()})
            grads.flat = gflat[0]
            mark("head forward + loss + backward")
            dx = out["grad_search_rows"]
            dz = self._new(B * 64, 256)
            self._check(self.lib.fear_nchw_to_nhwc(_p(out["grad_template"].contiguous()), _p(dz), B, 256, 64, 256, 0, st))
            if side is not None:
                side.wait_stream(main)                               # dz and the gradient buffer exist
                dz.record_stream(side)
                gall.record_stream(side)
                with torch.cuda.stream(side):
                    self._lane = 1
                    try:
                        fbwd(zctx, dz, gflat[1])
                    finally:
                        self._lane = 0          # (a raising kernel check must not leave later steps on the side lane's workspace)
                if self.mode == "block":
                    # third stream: the search pass's pointwise weight gradients, off the chain of input gradients
                    fbwd(xctx, dx, gflat[0], aux=self._aux)
                    mark("trunk backward: the search pass's chain of input gradients")
                    main.wait_stream(self._aux)
                    mark("  ... waiting for the weight-gradient stream")
                else:
                    fbwd(xctx, dx, gflat[0])
                main.wait_stream(side)
                mark("  ... waiting for the template pass")
            else:
                fbwd(xctx, dx, gflat[0])
                fbwd(zctx, dz, gflat[1])
            self._check(self.lib.fear_add(_p(gflat[0]), _p(gflat[1]), _p(gflat[0]), self._gtotal, st))   # shared parameters: the two passes add up (the trunk's slots come first)
            for L in self._trunk_layers():
                gw = self._gslot(gflat[0], L.conv_key, *L.w.shape)
                if L.kind == "dw":
                    grads[L.conv_key] = gw.t().reshape(L.cout, 1, L.k, L.k)
                elif L.kind == "stem":
                    grads[L.conv_key] = gw[:, :27].reshape(L.cout, 3, 3, 3)
                else:
                    grads[L.conv_key] = gw.reshape(L.cout, L.cin, 1, 1)
                grads[L.bn_key + ".weight"] = self._gslot(gflat[0], L.bn_key + ".weight", L.cout)
                grads[L.bn_key + ".bias"] = self._gslot(gflat[0], L.bn_key + ".bias", L.cout)
            grads.seal()
            mark("trunk backward: sum of the two passes, gradient views")
            if marks is not None:
                self.phase_marks = marks
            self.last_contexts = (zctx, xctx)          # saved activations of the two trunk passes (tests read the ReLU patterns)
        return out, grads

    allreduce_gradients = staticmethod(BoxTowerTrainHIP.allreduce_gradients)

    def parameter_slots(self) -> Dict[str, tuple]:
        """{parameter name: (storage tensor, to_storage, to_torch)} of every trainable parameter (see
        `BoxTowerTrainHIP.parameter_slots`): depthwise weights live as taps [k*k][C], the stem as [16][28] rows."""
        slots = {"connect_model." + k: v for k, v in self.head.parameter_slots().items()}
        layers = [self.stem, self.neck]
        for blk in self.blocks:
            layers += [L for L in (blk["pw"], blk["dw"], blk["pwl"]) if L is not None]
        for L in layers:
            N, K, k = L.cout, L.cin, L.k
            if L.kind == "dw":
                slots[L.conv_key] = (L.w, lambda g, N=N, k=k: g.reshape(N, k * k).t().contiguous(),
                                     lambda t, N=N, k=k: t.t().reshape(N, 1, k, k))
            elif L.kind == "stem":
                def pad28(g, N=N):
                    out = torch.zeros(N, 28, dtype=torch.float32, device=g.device)
                    out[:, :27] = g.reshape(N, 27)
                    return out
                slots[L.conv_key] = (L.w, pad28, lambda t, N=N: t[:, :27].reshape(N, 3, 3, 3).clone())
            else:
                slots[L.conv_key] = (L.w, lambda g, N=N, K=K: g.reshape(N, K).contiguous(), lambda t, N=N, K=K: t.reshape(N, K, 1, 1).clone())
            slots[L.bn_key + ".weight"] = (L.gamma, lambda g: g.contiguous(), lambda t: t.clone())
            slots[L.bn_key + ".bias"] = (L.beta, lambda g: g.contiguous(), lambda t: t.clone())
        return slots

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Current parameters and BatchNorm running statistics in the reference's layouts (CPU tensors, the keys of
        `random_init_state`): what `export.export_training_state` turns into inference weights."""
        out = {name: to_torch(t).cpu() for name, (t, _, to_torch) in self.parameter_slots().items()}
        out.update({k: v.cpu().clone() for k, v in self.running_stats().items()})
        return out

    def running_stats(self) -> Dict[str, torch.Tensor]:
        """{"<bn>.running_mean" / "<bn>.running_var": device tensor} of every BatchNorm, as updated by the `step` calls so far
        (template pass first, then the search pass, like two forward calls of the shared trunk) — what `export.py` folds."""
        out = {"connect_model." + k: v for k, v in self.head.running_stats().items()}
        layers = [self.stem, self.neck]
        for blk in self.blocks:
            layers += [L for L in (blk["pw"], blk["dw"], blk["pwl"]) if L is not None]
        for L in layers:
            out[L.bn_key + ".running_mean"] = L.running_mean
            out[L.bn_key + ".running_var"] = L.running_var
        return out

    def relu_patterns(self) -> Dict[str, List[torch.Tensor]]:
        """{conv name: [template-pass mask, search-pass mask]} (bool, NCHW, CPU) of every ReLU of the last `step`'s trunk passes,
        and {head layer prefix: [mask]} for the head — which elements the forward treated as active."""
        out: Dict[str, List[torch.Tensor]] = {}
        for ctx in self.last_contexts:
            saved = ctx[0]
            for rec in saved:
                if isinstance(rec, dict):                # fused unit: the mask is what every consumer recomputes, fma(pre, a, b) > 0
                    L, B, H = rec["L"], rec["B"], rec["H"]
                    if not L.relu:
                        continue
                    a, b, _ = rec["act"]
                    pre = rec["pre"]
                    if isinstance(pre, tuple):               # virtual expansion: (block input rows, W1)
                        pre = pre[0] @ pre[1].t()
                    y = torch.from_numpy(np.float32(pre.cpu().numpy().astype(np.float64) * a.cpu().numpy().astype(np.float64)
                                                    + b.cpu().numpy().astype(np.float64)))      # fp32 fma: exact product, one rounding
                    act = y
                else:
                    L, x, pre, act, mean, rstd, B, H = rec[:8]
                    if not L.relu:
                        continue
                    act = act.cpu()
                Ho = H // L.stride if L.kind == "dw" else H
                out.setdefault(L.name, []).append((act > 0).reshape(B, Ho, Ho, L.cout).permute(0, 3, 1, 2))
        for bname, br in self.head.branches.items():
            for L in [br["enc"], br["corr"]] + br["tower"]:
                m = (L.y.reshape(-1, L.ldy)[:, : L.cout] > 0)
                Bn = m.shape[0] // 256
                out[L.bn_prefix] = [m.reshape(Bn, 16, 16, L.cout).permute(0, 3, 1, 2).cpu()]
        return out
