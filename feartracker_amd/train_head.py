"""Training step of the FEAR correlation head on MI355X (SURVEY.md §8f N3 — first slice of BASELINE.json configs[4]).

`BoxTowerTrainHIP` is the training-mode counterpart of the reference's `BoxTower` (model_training/model/blocks.py:129-194)
followed by `FEARLoss` (model_training/train/loss.py:45-96): forward with BatchNorm on batch statistics, the two losses,
and the gradient of every parameter and of both inputs — what `FEARLightningModel._training_step` + `loss.backward()` do for
this part of the network (train/fear_lightning_model.py:60-66).  Every FLOP runs in the hand-written HIP operators of
include/fear_train.h (MFMA GEMMs for the pointwise convs' forward / dgrad / wgrad and the pixel-wise correlation and its two
gradients, fixed-order two-stage reductions for BatchNorm, bias and depthwise-weight gradients and the loss); this module
only sequences them and keeps the parameters, like the reference's Python does around torch.  torch is the allocator, the
stream provider and — for several ranks — the RCCL all-reduce of the flat gradient buffer (`allreduce_gradients`; the
reference uses Lightning DDP, train/trainer.py:50-52).  Parameter and gradient names are the reference's `state_dict` keys.

The trunk, SyncBatchNorm over the data-parallel group and the optimiser build on this module: `train_net.FEARNetTrainHIP`
(whole network, fused conv + BatchNorm operators), `SyncBN` below, `optim.AdamHIP`.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import numpy as np
import torch

from .hip_backend import load_library

_P = ctypes.c_void_p
_i, _l, _f, _d, _sz = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double, ctypes.c_size_t

TRAIN_SYMBOLS = {
    "fear_train_workspace_bytes": ([_l, _i], _sz),
    "fear_pw_forward": ([_P, _i, _P, _P, _P, _i, _l, _i, _i, _P], _i),
    "fear_pw_backward_data": ([_P, _i, _P, _P, _i, _P, _i, _l, _i, _i, _P], _i),
    "fear_pw_backward_weight": ([_P, _i, _P, _i, _P, _P, _sz, _l, _i, _i, _P], _i),
    "fear_col_sum": ([_P, _i, _P, _P, _sz, _l, _i, _P], _i),
    "fear_dw_forward": ([_P, _i, _P, _P, _P, _i, _i, _i, _i, _i, _i, _i, _P], _i),
    "fear_dw_backward_data": ([_P, _i, _P, _P, _i, _i, _i, _i, _i, _i, _i, _P], _i),
    "fear_dw_backward_weight": ([_P, _i, _P, _i, _P, _P, _sz, _i, _i, _i, _i, _i, _i, _P], _i),
    "fear_stem_im2col": ([_P, _P, _l, _i, _i, _P], _i),
    "fear_bn_train_forward": ([_P, _i, _P, _P, _P, _i, _P, _P, _P, _P, _d, _d, _l, _i, _i, _P, _sz, _P], _i),
    "fear_bn_train_backward": ([_P, _i, _P, _i, _P, _i, _P, _P, _P, _P, _i, _P, _P, _l, _i, _P, _sz, _P], _i),
    "fear_bn_reduce": ([_P, _i, _P, _l, _i, _P, _sz, _P], _i),
    "fear_bn_forward_from_sums": ([_P, _i, _P, _d, _P, _P, _P, _i, _P, _P, _P, _P, _d, _d, _l, _i, _i, _P], _i),
    "fear_bn_backward_reduce": ([_P, _i, _P, _i, _P, _i, _P, _P, _P, _l, _i, _P, _sz, _P], _i),
    "fear_bn_backward_from_sums": ([_P, _i, _P, _i, _P, _i, _P, _P, _P, _P, _d, _P, _P, _i, _P, _P, _P, _sz, _l, _i, _P], _i),
    "fear_pw_forward_stats": ([_P, _i, _P, _P, _i, _P, _P, _i, _l, _i, _i, _P, _P, _sz, _P], _i),
    "fear_dw_forward_stats": ([_P, _i, _P, _P, _i, _P, _P, _i, _i, _i, _i, _i, _i, _i, _P, _P, _sz, _P], _i),
    "fear_train_stats_workspace_bytes": ([_l, _i], _sz),
    "fear_bn_finalize": ([_P, _d, _P, _P, _P, _P, _P, _P, _P, _P, _d, _d, _i, _P], _i),
    "fear_bn_act": ([_P, _i, _P, _P, _i, _P, _i, _P, _i, _l, _i, _P], _i),
    "fear_bn_backward_reduce_x": ([_P, _i, _P, _i, _P, _P, _i, _P, _P, _P, _l, _i, _P, _sz, _P], _i),
    "fear_bn_backward_apply_x": ([_P, _i, _P, _i, _P, _P, _i, _P, _P, _P, _P, _d, _P, _P, _i, _P, _P, _P, _sz, _l, _i, _P], _i),
    "fear_bn_train_forward_ab": ([_P, _i, _P, _P, _i, _P, _i, _P, _i, _P, _P, _P, _P, _P, _P, _d, _d, _l, _i, _P, _sz, _P], _i),
    "fear_bn_train_backward_x": ([_P, _i, _P, _i, _P, _P, _i, _P, _P, _P, _P, _i, _P, _P, _l, _i, _P, _sz, _P], _i),
    "fear_pw_backward_weight_act": ([_P, _i, _P, _i, _P, _P, _i, _P, _P, _sz, _l, _i, _i, _P], _i),
    "fear_dw_backward_weight_act": ([_P, _i, _P, _i, _P, _P, _i, _P, _P, _sz, _i, _i, _i, _i, _i, _i, _P], _i),
    "fear_xcorr_forward": ([_P, _i, _P, _P, _i, _i, _i, _i, _i, _P], _i),
    "fear_xcorr_backward": ([_P, _i, _P, _i, _P, _P, _i, _P, _i, _P, _i, _i, _i, _i, _P], _i),
    "fear_exp_head_forward": ([_P, _P, _P, _P, _l, _P], _i),
    "fear_exp_head_backward": ([_P, _P, _P, _P, _P, _P, _P, _P, _sz, _l, _P], _i),
    "fear_head_loss": ([_P, _P, _P, _P, _P, _f, _f, _P, _P, _P, _P, _sz, _l, _P], _i),
    "fear_nchw_to_nhwc": ([_P, _P, _l, _i, _i, _i, _i, _P], _i),
    "fear_nhwc_to_nchw": ([_P, _P, _l, _i, _i, _i, _i, _P], _i),
    "fear_scale_column": ([_P, _i, _i, _f, _P, _i, _i, _l, _P], _i),
    "fear_add": ([_P, _P, _P, _l, _P], _i),
    "fear_adam_step": ([_P, _P, _P, _P, _l, _d, _d, _d, _d, _d, _i, _P], _i),
    # block-fused trunk operators (structs below mirror include/fear_train.h)
    "fear_irb_workspace_bytes": ([_P, _i, _i, _i], _sz),
    "fear_irb_scratch_floats": ([_P, _i, _i, _i], _sz),
    "fear_irb_virtual_ok": ([_P], _i),
    "fear_irb_train_forward": ([_P, _P, _P, _P, _i, _i, _i, _d, _d, _P, _sz, _P], _i),
    "fear_irb_train_backward": ([_P, _P, _P, _P, _P, _P, _P, _i, _i, _i, _P, _sz, _P, _P], _i),
    "fear_bn_running_update": ([_P, _d, _P, _P, _d, _d, _i, _P], _i),
    "fear_bn_running_update_multi": ([_P, _i, _d, _d, _P], _i),
    "fear_pwbn_workspace_bytes": ([_l, _i, _i], _sz),
    "fear_pwbn_train_forward": ([_P, _i, _P, _P, _P, _P, _P, _P, _P, _i, _P, _l, _i, _i, _d, _d, _P, _sz, _P], _i),
    "fear_pwbn_train_backward": ([_P, _P, _P, _i, _P, _i, _P, _P, _P, _P, _P, _P, _l, _i, _i, _P, _sz, _P, _P], _i),
    "fear_stem_workspace_bytes": ([_l, _i, _i], _sz),
    "fear_stem_train_forward": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _l, _i, _i, _d, _d, _P, _sz, _P], _i),
    "fear_stem_train_backward": ([_P, _P, _P, _P, _P, _P, _P, _P, _l, _i, _i, _P, _sz, _P, _P], _i),
    # the head's SepConv + BatchNorm + ReLU layer, one call per direction
    "fear_sepbn_workspace_bytes": ([_P, _i, _i, _i], _sz),
    "fear_sepbn_train_forward": ([_P, _P, _i, _P, _P, _P, _P, _i, _i, _i, _i, _d, _d, _P, _sz, _P], _i),
    "fear_sepbn_train_backward": ([_P, _P, _P, _i, _P, _P, _P, _P, _P, _P, _P, _i, _i, _i, _P, _sz, _P, _P], _i),
    # SyncBatchNorm hook of the block-fused operators: (stream, FearSync*)
    "fear_train_sync_bind": ([_P, _P], _i),
}


class FearIrbBlock(ctypes.Structure):
    """include/fear_train.h: one inverted-residual block's shape and parameters (device pointers, kernel layouts)."""
    _fields_ = [("cin", _i), ("cexp", _i), ("cout", _i), ("k", _i), ("stride", _i), ("expand", _i), ("residual", _i), ("flags", _i),
                ("w_pw", _P), ("w_dw", _P), ("w_pwl", _P), ("gamma", _P * 3), ("beta", _P * 3), ("running_mean", _P * 3), ("running_var", _P * 3)]


class FearIrbSaved(ctypes.Structure):
    _fields_ = [("e", _P), ("d", _P), ("p", _P), ("vec", _P * 3)]


class FearIrbGrads(ctypes.Structure):
    _fields_ = [("w_pw", _P), ("w_dw", _P), ("w_pwl", _P), ("gamma", _P * 3), ("beta", _P * 3)]


class FearBnRunning(ctypes.Structure):
    _fields_ = [("vec", _P), ("running_mean", _P), ("running_var", _P), ("C", _i), ("count", _d)]


class FearSepLayer(ctypes.Structure):
    """include/fear_train.h: one SepConv + BatchNorm + ReLU layer of the head (device pointers, kernel layouts)."""
    _fields_ = [("cin", _i), ("cout", _i), ("w_dw", _P), ("b_dw", _P), ("w_pw", _P), ("b_pw", _P), ("gamma", _P), ("beta", _P),
                ("running_mean", _P), ("running_var", _P)]


class FearSepGrads(ctypes.Structure):
    _fields_ = [("w_dw", _P), ("w_pw", _P), ("gamma", _P), ("beta", _P)]


_ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p)
FEAR_SYNC_BUF_BYTES = 16384


class FearSync(ctypes.Structure):
    """include/fear_train.h: the all-reduce hook a stream is bound to (fear_train_sync_bind)."""
    _fields_ = [("all_reduce", _ALLREDUCE_FN), ("user", _P), ("buf", _P), ("buf_bytes", _sz), ("world", _i)]


class GradDict(dict):
    """{parameter name: gradient in the reference's layout} whose values are views of ONE flat buffer in the kernels' storage
    layout (`flat`, the layout of the network's `param_flat`): the all-reduce of several ranks and the optimiser work on `flat`
    — one collective, one Adam launch — and the views follow.
    Contract: `flat` is the truth only while the dict is what the step made it.  Most entries are views of `flat` (in-place edits
    — `grads[k].mul_(c)`, the all-reduce — reach it); the depthwise and stem entries are re-laid-out COPIES.  `seal()` (called by
    the step) records the state; `current_flat()` returns `flat` unless an entry was rebound (`grads[k] = grads[k] * c`, as
    gradient clipping does) or a copy entry was edited in place — then None: `optim.AdamHIP.step` and `allreduce_gradients` fall
    back to the dict's values, so such an edit is never silently dropped."""
    flat: Optional[torch.Tensor] = None
    _sealed = None
    _rebound = False

    def seal(self) -> "GradDict":
        base = self.flat.untyped_storage().data_ptr() if self.flat is not None else None
        self._sealed = {k: v._version for k, v in self.items() if base is None or v.untyped_storage().data_ptr() != base}
        self._rebound = False
        return self

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        if self._sealed is not None:
            self._rebound = True

    def __delitem__(self, key):
        super().__delitem__(key)
        if self._sealed is not None:
            self._rebound = True

    def current_flat(self) -> Optional[torch.Tensor]:
        if self.flat is None or self._sealed is None:
            return self.flat
        if self._rebound or any(k not in self or self[k]._version != ver for k, ver in self._sealed.items()):
            return None
        return self.flat

_bound = None


def load_train_library() -> ctypes.CDLL:
    """The training operators live in the same libfear_hip.so; declare their prototypes (include/fear_train.h)."""
    global _bound
    if _bound is None:
        lib = load_library()
        for name, (args, res) in TRAIN_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, res
        _bound = lib
    return _bound


class TrainError(RuntimeError):
    pass


def _p(t: Optional[torch.Tensor], offset: int = 0):
    return None if t is None else ctypes.c_void_p(t.data_ptr() + 4 * offset)


class SyncBN:
    """SyncBatchNorm across the ranks of a process group (the reference's multi-GPU backends set `sync_bn: True`,
    config/backend/{2,4}gpu.yaml): every BatchNorm adds its float64 sums over the ranks with one all-reduce in the forward
    and one in the backward (include/fear_train.h).  Every rank must feed the same number of rows (DDP's equal batches)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)

    def all_reduce(self, sums: torch.Tensor) -> None:
        self.dist.all_reduce(sums, op=self.dist.ReduceOp.SUM, group=self.group)


class SyncHook:
    """SyncBatchNorm for the block-fused operators (`fear_irb_train_*`, `fear_pwbn_*`, `fear_stem_train_*`, `fear_sepbn_*`): their
    BatchNorm reductions sit inside one C call, so the ranks' all-reduce is a callback the library makes between a producer's
    float64 sums and their finalize (include/fear_train.h, `fear_train_sync_bind`).  `bound(*torch streams)` binds each stream to a
    buffer of its own for the duration of a step; the callback runs the group's all-reduce (`sync.all_reduce`: RCCL through
    torch.distributed) on that stream's buffer, ordered on that stream.  Collectives of several streams reach the communicator in
    the host's issue order — the same on every rank, because every rank runs the same Python.  `sync` is a `SyncBN` (or anything
    with `.world` and `.all_reduce(tensor)`: the tests play two ranks on one GPU with it)."""

    def __init__(self, lib, sync, device):
        self.lib, self.sync, self.device = lib, sync, device
        self._slots = {}                      # raw stream handle -> [torch stream, buffer, FearSync, depth]
        self.error = None
        self._cb = _ALLREDUCE_FN(self._all_reduce)      # (kept alive: the library holds the pointer while a stream is bound)

    def _all_reduce(self, user, buf, n, is_f32, stream):
        try:
            stream_t, buffer = self._slots[int(stream or 0)][:2]
            view = buffer.view(torch.float32)[:n] if is_f32 else buffer[:n]
            with torch.cuda.stream(stream_t):
                self.sync.all_reduce(view)
            return 0
        except BaseException as exc:          # noqa: BLE001 - must not unwind through the C frames; re-raised by the caller's _check
            self.error = exc
            return 1

    def bound(self, *streams):
        hook = self

        class _Bound:
            def __enter__(self_inner):
                self_inner.handles = []
                for st in streams:
                    if st is None:
                        continue
                    h = int(st.cuda_stream)
                    slot = hook._slots.get(h)
                    if slot is None:
                        buf = torch.zeros(FEAR_SYNC_BUF_BYTES // 8, dtype=torch.float64, device=hook.device)
                        fs = FearSync(hook._cb, None, buf.data_ptr(), FEAR_SYNC_BUF_BYTES, int(hook.sync.world))
                        slot = hook._slots[h] = [st, buf, fs, 0]
                    if slot[3] == 0:
                        rc = hook.lib.fear_train_sync_bind(ctypes.c_void_p(h), ctypes.byref(slot[2]))
                        if rc != 0:
                            raise TrainError(f"fear_train_sync_bind failed with status {rc}")
                    slot[3] += 1
                    self_inner.handles.append(h)
                return hook

            def __exit__(self_inner, *exc):
                for h in self_inner.handles:
                    slot = hook._slots[h]
                    slot[3] -= 1
                    if slot[3] == 0:
                        hook.lib.fear_train_sync_bind(ctypes.c_void_p(h), None)
                return False
        return _Bound()


def bn_forward(lib, st, ws, wsb, sync: Optional["SyncBN"], x, ldx, gamma, beta, out, ld_out, mean, rstd, running_mean, running_var,
               momentum, eps, M, C, relu) -> int:
    if sync is None:
        return lib.fear_bn_train_forward(_p(x), ldx, _p(gamma), _p(beta), _p(out), ld_out, _p(mean), _p(rstd), _p(running_mean),
                                         _p(running_var), momentum, eps, M, C, relu, ws, wsb, st)
    sums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
    rc = lib.fear_bn_reduce(_p(x), ldx, _p(sums), M, C, ws, wsb, st)
    if rc != 0:
        return rc
    sync.all_reduce(sums)
    return lib.fear_bn_forward_from_sums(_p(x), ldx, _p(sums), float(M) * sync.world, _p(gamma), _p(beta), _p(out), ld_out, _p(mean),
                                         _p(rstd), _p(running_mean), _p(running_var), momentum, eps, M, C, relu, st)


def bn_backward(lib, st, ws, wsb, sync: Optional["SyncBN"], dy, lddy, y_act, ldy, x, ldx, mean, rstd, gamma, dx, lddx, dgamma, dbeta,
                M, C) -> int:
    if sync is None:
        return lib.fear_bn_train_backward(_p(dy), lddy, _p(y_act), ldy, _p(x), ldx, _p(mean), _p(rstd), _p(gamma), _p(dx), lddx,
                                          _p(dgamma), _p(dbeta), M, C, ws, wsb, st)
    sums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
    rc = lib.fear_bn_backward_reduce(_p(dy), lddy, _p(y_act), ldy, _p(x), ldx, _p(mean), _p(rstd), _p(sums), M, C, ws, wsb, st)
    if rc != 0:
        return rc
    local = sums.clone()
    sync.all_reduce(sums)
    return lib.fear_bn_backward_from_sums(_p(dy), lddy, _p(y_act), ldy, _p(x), ldx, _p(mean), _p(rstd), _p(gamma), _p(sums),
                                          float(M) * sync.world, _p(local), _p(dx), lddx, _p(dgamma), _p(dbeta), ws, wsb, M, C, st)


class _Sep:
    """One SepConv (+ optional BatchNorm + ReLU) with its parameters in kernel layout and its saved activations."""

    def __init__(self, owner, prefix: str, bn_prefix: Optional[str], sd: Dict[str, torch.Tensor], pad_out_to: int = 0):
        dev = owner.device
        dw = sd[prefix + ".depthwise.weight"].float()
        pw = sd[prefix + ".pointwise.weight"].float()
        self.prefix, self.bn_prefix = prefix, bn_prefix
        self.cin, self.cout = dw.shape[0], pw.shape[0]
        self.n = max(self.cout, pad_out_to)                     # pointwise rows padded to a multiple of 4 (cls_pred: 1 -> 4)
        self.taps = dw.reshape(self.cin, 9).t().contiguous().to(dev)                       # [9][C]
        self.dw_bias = sd[prefix + ".depthwise.bias"].float().to(dev) if prefix + ".depthwise.bias" in sd else None
        w = torch.zeros(self.n, self.cin)
        w[: self.cout] = pw.reshape(self.cout, self.cin)
        self.w = w.to(dev)
        self.pw_bias = None
        if prefix + ".pointwise.bias" in sd:
            b = torch.zeros(self.n)
            b[: self.cout] = sd[prefix + ".pointwise.bias"].float()
            self.pw_bias = b.to(dev)
        if bn_prefix:
            self.gamma = sd[bn_prefix + ".weight"].float().to(dev)
            self.beta = sd[bn_prefix + ".bias"].float().to(dev)
            self.running_mean = sd[bn_prefix + ".running_mean"].float().clone().to(dev)
            self.running_var = sd[bn_prefix + ".running_var"].float().clone().to(dev)


class BoxTowerTrainHIP:
    """BoxTower(towernum=2, inchannels=256, outchannels=256, mobile=True) in training mode + FEARLoss, on HIP operators."""

    S, TZ = 16, 8           # search feature map 16x16, template feature map 8x8 (256 / 128 px crops, stride 16)

    def __init__(self, state_dict: Dict[str, "np.ndarray | torch.Tensor"], device: int = 0, momentum: float = 0.1,
                 eps: float = 1e-5, coef_cls: float = 1.0, coef_reg: float = 1.0, sync_bn: bool = False, group=None,
                 fused: bool = True):
        """`fused` (default): every SepConv + BatchNorm + ReLU layer is one call per direction (`fear_sepbn_train_*`: statistics in
        the GEMM's epilogue, the BatchNorm backward formed on load, weight gradients on `aux_stream`) — with SyncBatchNorm the
        ranks' all-reduces are made by the library's hook between a layer's sums and their finalize (`SyncHook`); False: one
        operator per pass, as rounds 2-4 ran it (SyncBatchNorm: its collectives between the passes).
        `sync_bn`: True (a `SyncBN` over `group`) or an object with `.world` and `.all_reduce(tensor)`."""
        if not torch.cuda.is_available():
            raise RuntimeError("BoxTowerTrainHIP needs a ROCm GPU; there is no CPU fallback")
        self.fused = bool(fused)
        self.lib = load_train_library()
        self.sync = (SyncBN(group) if sync_bn is True else sync_bn) if sync_bn else None
        self.device = torch.device(f"cuda:{int(device)}")
        self.momentum, self.eps, self.coef_cls, self.coef_reg = momentum, eps, coef_cls, coef_reg
        sd = {k: torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v.detach().cpu() for k, v in state_dict.items()}
        self.branches = {}
        for name, enc, corr, tower, pred in (("cls", "cls_encode.matrix11_s", "cls_dw.enc", "cls_tower", "cls_pred"),
                                             ("reg", "reg_encode.matrix11_s", "reg_dw.enc", "bbox_tower", "bbox_pred")):
            self.branches[name] = dict(
                enc=_Sep(self, enc + ".0", enc + ".1", sd), corr=_Sep(self, corr + ".0", corr + ".1", sd),
                tower=[_Sep(self, f"{tower}.0", f"{tower}.1", sd), _Sep(self, f"{tower}.3", f"{tower}.4", sd)],
                pred=_Sep(self, pred, None, sd, pad_out_to=4))
        self.adjust = sd["adjust"].float().reshape(1).to(self.device)
        self.bias4 = sd["bias"].float().reshape(4).to(self.device)
        self.hook = SyncHook(self.lib, self.sync, self.device) if (self.sync is not None and self.fused) else None
        self._ws = {}                # per stream lane
        self._lane = 0
        self._galloc = None          # set by FEARNetTrainHIP: gradient tensors are views of its flat gradient buffer
        # a second HIP stream for the regression branch: the two towers are independent between the shared input features and the
        # loss (forward) and between the loss gradient and the sum of their input gradients (backward), and every kernel of a 16 x 16
        # map is a 256-workgroup launch that leaves most of a 256-CU device idle.  Not with SyncBatchNorm (its collectives must be
        # issued in one order on every rank).  Set by FEARNetTrainHIP (two_streams); None = one stream.
        self.side_stream = None
        # a stream for the weight gradients of the SepConv layers (they feed nothing in the backward pass: the chain of input
        # gradients — four 256-workgroup kernels a layer — runs on without them).  `aux_join`: wait for it at the end of step();
        # FEARNetTrainHIP shares the stream with its trunk and joins once, after the trunk's backward.  None = in line.
        self.aux_stream = None
        self.aux_join = True

    def _layers(self):
        for br in self.branches.values():
            for L in [br["enc"], br["corr"]] + br["tower"] + [br["pred"]]:
                yield L

    def rehome_parameters(self, alloc) -> None:
        """Move every parameter into storage handed out by `alloc(name, tensor) -> tensor of the same shape holding the same
        values` (FEARNetTrainHIP: views of one flat buffer, so that the optimiser is one launch); names as `parameter_slots`."""
        for L in self._layers():
            L.desc = None
            L.taps = alloc(L.prefix + ".depthwise.weight", L.taps)
            if L.dw_bias is not None:
                L.dw_bias = alloc(L.prefix + ".depthwise.bias", L.dw_bias)
            L.w = alloc(L.prefix + ".pointwise.weight", L.w)
            if L.pw_bias is not None:
                L.pw_bias = alloc(L.prefix + ".pointwise.bias", L.pw_bias)
            if L.bn_prefix:
                L.gamma = alloc(L.bn_prefix + ".weight", L.gamma)
                L.beta = alloc(L.bn_prefix + ".bias", L.beta)
        self.adjust = alloc("adjust", self.adjust)
        self.bias4 = alloc("bias", self.bias4)

    def _gnew(self, name: str, *shape) -> torch.Tensor:
        """Storage of the gradient of parameter `name` (kernel layout)."""
        return self._galloc(name, *shape) if self._galloc is not None else self._new(*shape)

    # ------------------------------------------------------------------ plumbing
    def _check(self, st: int) -> None:
        if st != 0:
            err = self.hook.error if self.hook is not None else None
            if st == -8 and err is not None:            # FEAR_TRAIN_ERR_SYNC: the all-reduce callback raised
                self.hook.error = None
                raise TrainError("the SyncBatchNorm all-reduce of a block-fused operator failed") from err
            raise TrainError(f"libfear_hip training operator failed with status {st}")

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _new(self, *shape) -> torch.Tensor:
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _workspace(self, rows: int):
        need = int(self.lib.fear_train_workspace_bytes(rows, 320))
        need = max(need, (8 * rows + rows // 8 + 4096) * 4)
        ws = self._ws.get(self._lane)
        if ws is None or ws.numel() * 4 < need:
            self._ws[self._lane] = None
            ws = self._ws[self._lane] = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.device)
        return _p(ws), ws.numel() * 4

    # ------------------------------------------------------------------ one SepConv [+ BN + ReLU]
    def _sep_desc(self, L: _Sep) -> "FearSepLayer":
        if getattr(L, "desc", None) is None:
            ptr = lambda t: None if t is None else t.data_ptr()
            L.desc = FearSepLayer(L.cin, L.cout, ptr(L.taps), ptr(L.dw_bias), ptr(L.w), ptr(L.pw_bias), ptr(L.gamma), ptr(L.beta),
                                  ptr(L.running_mean), ptr(L.running_var))
        return L.desc

    def _sep_workspace(self, L: _Sep, B: int):
        """The workspace of the one-call SepConv + BN layers of this stream lane (its weight-gradient regions belong to the
        weight-gradient stream: kept apart from the lane's other workspace)."""
        need = int(self.lib.fear_sepbn_workspace_bytes(ctypes.byref(self._sep_desc(L)), B, self.S, self.S))
        if need == 0:
            raise TrainError(f"{L.prefix}: shape not supported by fear_sepbn_train_*")
        key = ("sep", self._lane)
        ws = self._ws.get(key)
        if ws is None or ws.numel() * 4 < need:
            self._ws[key] = None
            ws = self._ws[key] = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.device)
        return _p(ws), ws.numel() * 4

    def _sep_forward(self, L: _Sep, x: torch.Tensor, ldx: int, B: int, out: Optional[torch.Tensor] = None, ld_out: int = 0):
        lib, st, M = self.lib, self._stream(), B * self.S * self.S
        if L.bn_prefix and self.fused:
            # one call: depthwise, pointwise with the statistics in its epilogue, finalize, activation (csrc/fear_train_block.h)
            ws, wsb = self._sep_workspace(L, B)
            L.x, L.ldx = x, ldx
            L.d, L.p, L.vec = self._new(M, L.cin), self._new(M, L.cout), self._new(4 * L.cout)
            if out is None:
                out, ld_out = self._new(M, L.cout), L.cout
            self._check(lib.fear_sepbn_train_forward(ctypes.byref(self._sep_desc(L)), _p(x), ldx, _p(L.d), _p(L.p), _p(L.vec), _p(out),
                                                     ld_out, B, self.S, self.S, self.momentum, self.eps, ws, wsb, st))
            L.y, L.ldy = out, ld_out                   # (FEARNetTrainHIP.relu_patterns reads the activation)
            return out
        ws, wsb = self._workspace(M)
        L.x, L.ldx = x, ldx
        L.d = self._new(M, L.cin)
        self._check(lib.fear_dw_forward(_p(x), ldx, _p(L.taps), _p(L.dw_bias), _p(L.d), L.cin, B, self.S, self.S, L.cin, 3, 1, st))
        L.p = self._new(M, L.n)
        self._check(lib.fear_pw_forward(_p(L.d), L.cin, _p(L.w), _p(L.pw_bias), _p(L.p), L.n, M, L.cin, L.n, st))
        if not L.bn_prefix:
            return L.p
        L.mean, L.rstd = self._new(L.cout), self._new(L.cout)
        if out is None:
            out, ld_out = self._new(M, L.cout), L.cout
        L.y, L.ldy = out, ld_out
        self._check(bn_forward(lib, st, ws, wsb, self.sync, L.p, L.n, L.gamma, L.beta, out, ld_out, L.mean, L.rstd,
                               L.running_mean, L.running_var, self.momentum, self.eps, M, L.cout, 1))
        return out

    def _sep_backward(self, L: _Sep, dy: torch.Tensor, lddy: int, B: int, grads: Dict[str, torch.Tensor]) -> torch.Tensor:
        """dy = gradient w.r.t. the layer's output (after BN+ReLU when it has them); returns d(input) as [M][cin]."""
        lib, st, M = self.lib, self._stream(), B * self.S * self.S
        if L.bn_prefix and self.fused:
            if lddy != L.cout:
                raise TrainError(f"{L.prefix}: fear_sepbn_train_backward takes the output gradient as contiguous rows of {L.cout}, got pitch {lddy}")
            ws, wsb = self._sep_workspace(L, B)
            dd, coef, dx = self._new(M, L.cin), self._new(4 * L.cout), self._new(M, L.cin)
            dtaps, dw = self._gnew(L.prefix + ".depthwise.weight", 9, L.cin), self._gnew(L.prefix + ".pointwise.weight", L.n, L.cin)
            dgamma, dbeta = self._gnew(L.bn_prefix + ".weight", L.cout), self._gnew(L.bn_prefix + ".bias", L.cout)
            gr = FearSepGrads(dtaps.data_ptr(), dw.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr())
            aux = self.aux_stream
            self._check(lib.fear_sepbn_train_backward(ctypes.byref(self._sep_desc(L)), ctypes.byref(gr), _p(L.x), L.ldx, _p(L.d), _p(L.p),
                                                      _p(L.vec), _p(dy), _p(dd), _p(coef), _p(dx), B, self.S, self.S, ws, wsb, st,
                                                      ctypes.c_void_p(aux.cuda_stream) if aux is not None else None))
            if aux is not None:
                for t in (dy, dd, coef, L.x, L.d, L.p, L.vec, dtaps, dw):
                    t.record_stream(aux)
            grads[L.bn_prefix + ".weight"], grads[L.bn_prefix + ".bias"] = dgamma, dbeta
            grads[L.prefix + ".pointwise.weight"] = dw[: L.cout].reshape(L.cout, L.cin, 1, 1)
            grads[L.prefix + ".depthwise.weight"] = dtaps.t().reshape(L.cin, 1, 3, 3)
            # (biases in front of a BatchNorm: exactly-zero gradients, see below)
            for key, bias, n in ((".pointwise.bias", L.pw_bias, L.n), (".depthwise.bias", L.dw_bias, L.cin)):
                if bias is not None:
                    db = self._gnew(L.prefix + key, n)
                    if self._galloc is None:
                        db.zero_()
                    grads[L.prefix + key] = db[: L.cout] if key == ".pointwise.bias" else db
            return dx
        ws, wsb = self._workspace(M)
        if L.bn_prefix:
            dp = self._new(M, L.n)
            dgamma, dbeta = self._gnew(L.bn_prefix + ".weight", L.cout), self._gnew(L.bn_prefix + ".bias", L.cout)
            self._check(bn_backward(lib, st, ws, wsb, self.sync, dy, lddy, L.y, L.ldy, L.p, L.n, L.mean, L.rstd, L.gamma, dp, L.n,
                                    dgamma, dbeta, M, L.cout))
            grads[L.bn_prefix + ".weight"], grads[L.bn_prefix + ".bias"] = dgamma, dbeta
            lddp = L.n
        else:
            dp, lddp = dy, lddy
        dw = self._gnew(L.prefix + ".pointwise.weight", L.n, L.cin)
        self._check(lib.fear_pw_backward_weight(_p(dp), lddp, _p(L.d), L.cin, _p(dw), ws, wsb, M, L.cin, L.n, st))
        grads[L.prefix + ".pointwise.weight"] = dw[: L.cout].reshape(L.cout, L.cin, 1, 1)
        # A bias in front of a BatchNorm has an exactly-zero gradient (the normalisation removes any per-channel constant: the
        # column sums of d(pre) vanish, and with them those of the depthwise bias one conv further up); torch's autograd yields
        # rounding noise of 1e-9 there.  Those sums are not taken — two reductions and four launches per layer — the gradient is
        # the zero it is.  Only the prediction heads (no BatchNorm behind them) have bias gradients.
        zero_bias = L.bn_prefix is not None
        if L.pw_bias is not None:
            db = self._gnew(L.prefix + ".pointwise.bias", L.n)
            if zero_bias:
                if self._galloc is None:          # (FEARNetTrainHIP's gradient buffer starts from zeros)
                    db.zero_()
            else:
                self._check(lib.fear_col_sum(_p(dp), lddp, _p(db), ws, wsb, M, L.n, st))
            grads[L.prefix + ".pointwise.bias"] = db[: L.cout]
        dd = self._new(M, L.cin)
        self._check(lib.fear_pw_backward_data(_p(dp), lddp, _p(L.w), None, 0, _p(dd), L.cin, M, L.cin, L.n, st))
        dtaps = self._gnew(L.prefix + ".depthwise.weight", 9, L.cin)
        self._check(lib.fear_dw_backward_weight(_p(dd), L.cin, _p(L.x), L.ldx, _p(dtaps), ws, wsb, B, self.S, self.S, L.cin, 3, 1, st))
        grads[L.prefix + ".depthwise.weight"] = dtaps.t().reshape(L.cin, 1, 3, 3)
        if L.dw_bias is not None:
            dbd = self._gnew(L.prefix + ".depthwise.bias", L.cin)
            if zero_bias:
                if self._galloc is None:
                    dbd.zero_()
            else:
                self._check(lib.fear_col_sum(_p(dd), L.cin, _p(dbd), ws, wsb, M, L.cin, st))
            grads[L.prefix + ".depthwise.bias"] = dbd
        dx = self._new(M, L.cin)
        self._check(lib.fear_dw_backward_data(_p(dd), L.cin, _p(L.taps), _p(dx), L.cin, B, self.S, self.S, L.cin, 3, 1, st))
        return dx

    # ------------------------------------------------------------------ the step
    @torch.no_grad()
    def step(self, search_feats: torch.Tensor, template_feats: torch.Tensor, gt_reg: torch.Tensor, gt_cls: torch.Tensor,
             gt_weight: torch.Tensor) -> Dict[str, object]:
        """search_feats (B,256,16,16), template_feats (B,256,8,8) NCHW fp32 (the neck's outputs); gt_* as the reference's
        dataset emits them (siam_dataset.py:53-56): regression map (B,4,16,16), label (B,1,16,16), weight (B,16,16).
        Returns {"loss_cls", "loss_reg", "bbox", "cls", "grads": {reference parameter name: tensor}, "grad_search",
        "grad_template"}."""
        lib, dev = self.lib, self.device
        xs = search_feats.to(dev, torch.float32).contiguous()
        B = xs.shape[0]
        if tuple(xs.shape[1:]) != (256, self.S, self.S):
            raise ValueError("expected search features (B,256,16,16) and template features (B,256,8,8)")
        P = self.S * self.S
        with torch.cuda.device(dev):
            st = self._stream()
            x = self._new(B * P, 256)
            self._check(lib.fear_nchw_to_nhwc(_p(xs), _p(x), B, 256, P, 256, 0, st))
            out = self.step_rows(x, template_feats, gt_reg, gt_cls, gt_weight)
            grad_search = self._new(B, 256, self.S, self.S)
            self._check(lib.fear_nhwc_to_nchw(_p(out.pop("grad_search_rows")), _p(grad_search), B, 256, P, 256, 0, st))
        out["grad_search"] = grad_search
        return out

    @torch.no_grad()
    def step_rows(self, x: torch.Tensor, template_feats: torch.Tensor, gt_reg: torch.Tensor, gt_cls: torch.Tensor,
                  gt_weight: torch.Tensor) -> Dict[str, object]:
        """`_step_rows`, with this stream and the side stream bound to the SyncBatchNorm hook when the head runs its one-call layers
        in a data-parallel group (a binding the caller already holds is kept: `SyncHook.bound` counts)."""
        if self.hook is None:
            return self._step_rows(x, template_feats, gt_reg, gt_cls, gt_weight)
        with torch.cuda.device(self.device):
            with self.hook.bound(torch.cuda.current_stream(self.device), self.side_stream):
                return self._step_rows(x, template_feats, gt_reg, gt_cls, gt_weight)

    @torch.no_grad()
    def _step_rows(self, x: torch.Tensor, template_feats: torch.Tensor, gt_reg: torch.Tensor, gt_cls: torch.Tensor,
                   gt_weight: torch.Tensor) -> Dict[str, object]:
        """`step` on the search features as the trunk leaves them — pixel rows [B*256][256] (channels last) — returning
        "grad_search_rows" in the same layout instead of "grad_search": FEARNetTrainHIP's trunk works on rows on both sides of the
        head, and the two layout passes each way were 0.2 ms of its step.  The template features stay (B,256,8,8) NCHW: the
        correlation reads them as [B][256][64] matrices."""
        lib, dev = self.lib, self.device
        zs = template_feats.to(dev, torch.float32).contiguous()
        B = zs.shape[0]
        M, P, J = B * self.S * self.S, self.S * self.S, self.TZ * self.TZ
        if tuple(zs.shape) != (B, 256, self.TZ, self.TZ) or tuple(x.shape) != (M, 256) or not x.is_contiguous() or x.device != dev:
            raise ValueError("expected search feature rows (B*256,256) on the device and template features (B,256,8,8)")
        with torch.cuda.device(dev):
            st = self._stream()
            ws, wsb = self._workspace(M)
            saved = {}
            pred_out = {}
            main = torch.cuda.current_stream(dev)
            side = self.side_stream if (self.sync is None or self.hook is not None) else None

            def on_branch_streams(fn):
                """fn(name, branch) for both towers: "reg" on the side stream (lane 1), "cls" on this one; joined on return."""
                items = list(self.branches.items())
                if side is None:
                    for name, br in items:
                        fn(name, br)
                    return
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._lane = 1
                    try:
                        fn(*items[1])
                    finally:
                        self._lane = 0
                fn(*items[0])
                main.wait_stream(side)

            def forward_branch(name, br):
                st_ = self._stream()
                cat = self._new(M, 320)                                   # [encode output | correlation] without a concat kernel
                self._sep_forward(br["enc"], x, 256, B, out=cat, ld_out=320)
                self._check(lib.fear_xcorr_forward(_p(cat), 320, _p(zs), _p(cat, 256), 320, B, P, 256, J, st_))
                a = self._sep_forward(br["corr"], cat, 320, B)
                for L in br["tower"]:
                    a = self._sep_forward(L, a, 256, B)
                pred_out[name] = self._sep_forward(br["pred"], a, 256, B)   # [M][4] (cls: column 0 is real)
                saved[name] = cat
            if side is not None:
                x.record_stream(side)
                zs.record_stream(side)
            on_branch_streams(forward_branch)
            if side is not None:
                pred_out["reg"].record_stream(main)
            bbox_rows = self._new(M, 4)
            self._check(lib.fear_exp_head_forward(_p(pred_out["reg"]), _p(self.adjust), _p(self.bias4), _p(bbox_rows), M, st))
            cls_rows = self._new(M)
            self._check(lib.fear_scale_column(_p(pred_out["cls"]), 4, 0, 0.1, _p(cls_rows), 1, 0, M, st))   # cls = 0.1 * cls_pred(c)
            # ---- loss + its gradient
            gr = self._new(M, 4)
            self._check(lib.fear_nchw_to_nhwc(_p(gt_reg.to(dev, torch.float32).contiguous()), _p(gr), B, 4, P, 4, 0, st))
            gc = gt_cls.to(dev, torch.float32).contiguous().reshape(M)
            gw = gt_weight.to(dev, torch.float32).contiguous().reshape(M)
            losses, dbbox, dcls = self._new(2), self._new(M, 4), self._new(M)
            self._check(lib.fear_head_loss(_p(bbox_rows), _p(cls_rows), _p(gr), _p(gc), _p(gw), self.coef_cls, self.coef_reg,
                                           _p(losses), _p(dbbox), _p(dcls), ws, wsb, M, st))
            # ---- backward
            grads: Dict[str, torch.Tensor] = {}
            dpred = {}
            dp_reg, dadj, dbias = self._new(M, 4), self._gnew("adjust", 1), self._gnew("bias", 4)
            self._check(lib.fear_exp_head_backward(_p(pred_out["reg"]), _p(self.adjust), _p(bbox_rows), _p(dbbox), _p(dp_reg),
                                                   _p(dadj), _p(dbias), ws, wsb, M, st))
            grads["adjust"], grads["bias"] = dadj, dbias.reshape(1, 4, 1, 1)
            dpred["reg"] = dp_reg
            dp_cls = torch.zeros((M, 4), dtype=torch.float32, device=dev)
            self._check(lib.fear_scale_column(_p(dcls), 1, 0, 0.1, _p(dp_cls), 4, 0, M, st))
            dpred["cls"] = dp_cls
            dxz = {}

            def backward_branch(name, br):
                st_ = self._stream()
                da = self._sep_backward(br["pred"], dpred[name], 4, B, grads)
                for L in reversed(br["tower"]):
                    da = self._sep_backward(L, da, 256, B, grads)
                dcat = self._sep_backward(br["corr"], da, 256, B, grads)            # [M][320]
                cat = saved[name]
                denc, dz = self._new(M, 256), self._new(B, 256, self.TZ, self.TZ)
                self._check(lib.fear_xcorr_backward(_p(dcat, 256), 320, _p(cat), 320, _p(zs), _p(dcat), 320, _p(denc), 256, _p(dz),
                                                    B, P, 256, J, st_))
                dxz[name] = (self._sep_backward(br["enc"], denc, 256, B, grads), dz)
            if side is not None:
                dpred["reg"].record_stream(side)
            on_branch_streams(backward_branch)
            (dx_total, dz_total), (dx, dz) = dxz["cls"], dxz["reg"]
            if side is not None:
                dx.record_stream(main)
                dz.record_stream(main)
            self._check(lib.fear_add(_p(dx_total), _p(dx), _p(dx_total), dx.numel(), st))
            self._check(lib.fear_add(_p(dz_total), _p(dz), _p(dz_total), dz.numel(), st))
            bbox = self._new(B, 4, self.S, self.S)
            self._check(lib.fear_nhwc_to_nchw(_p(bbox_rows), _p(bbox), B, 4, P, 4, 0, st))
            if self.aux_stream is not None and self.aux_join:
                main.wait_stream(self.aux_stream)
        return {"loss_cls": losses[0], "loss_reg": losses[1], "bbox": bbox, "cls": cls_rows.reshape(B, 1, self.S, self.S),
                "grads": grads, "grad_search_rows": dx_total, "grad_template": dz_total}

    # ------------------------------------------------------------------ several ranks
    @staticmethod
    def allreduce_gradients(grads: Dict[str, torch.Tensor], group=None) -> Dict[str, torch.Tensor]:
        """Data-parallel gradient averaging as ONE collective: the gradients are flattened into one buffer (0.65 M floats for
        the head), all-reduced (RCCL over xGMI with the "nccl" backend; gloo in the CPU tests) and divided by the world size —
        what Lightning DDP does bucket by bucket for the reference (train/trainer.py:50-52)."""
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return grads
        flat = grads.current_flat() if isinstance(grads, GradDict) else getattr(grads, "flat", None)
        if flat is not None:                       # FEARNetTrainHIP: the gradients ARE one buffer already; its views follow
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            flat /= dist.get_world_size(group)
            return grads
        names = sorted(grads)
        flat = torch.cat([grads[n].reshape(-1) for n in names])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= dist.get_world_size(group)
        out, off = {}, 0
        for n in names:
            k = grads[n].numel()
            out[n] = flat[off: off + k].reshape(grads[n].shape)
            off += k
        return out

    def parameter_slots(self) -> Dict[str, tuple]:
        """{parameter name: (storage tensor, to_storage, to_torch)}: the device tensor the kernels read (kernel layout: depthwise
        taps [9][C], pointwise rows padded to a multiple of 4), a function mapping a gradient / value in the reference's
        layout onto it, and its inverse.  What `optim.AdamHIP` updates and `state_dict` reads."""
        slots: Dict[str, tuple] = {}

        def add_sep(L: _Sep):
            C, N, n = L.cin, L.cout, L.n
            slots[L.prefix + ".depthwise.weight"] = (L.taps, lambda g, C=C: g.reshape(C, 9).t().contiguous(),
                                                     lambda t, C=C: t.t().reshape(C, 1, 3, 3))
            if L.dw_bias is not None:
                slots[L.prefix + ".depthwise.bias"] = (L.dw_bias, lambda g: g.contiguous(), lambda t: t.clone())

            def pad_rows(g, N=N, n=n, C=C):
                out = torch.zeros(n, C, dtype=torch.float32, device=g.device)
                out[:N] = g.reshape(N, C)
                return out
            slots[L.prefix + ".pointwise.weight"] = (L.w, pad_rows, lambda t, N=N, C=C: t[:N].reshape(N, C, 1, 1).clone())
            if L.pw_bias is not None:
                def pad_vec(g, N=N, n=n):
                    out = torch.zeros(n, dtype=torch.float32, device=g.device)
                    out[:N] = g.reshape(N)
                    return out
                slots[L.prefix + ".pointwise.bias"] = (L.pw_bias, pad_vec, lambda t, N=N: t[:N].clone())
            if L.bn_prefix:
                slots[L.bn_prefix + ".weight"] = (L.gamma, lambda g: g.contiguous(), lambda t: t.clone())
                slots[L.bn_prefix + ".bias"] = (L.beta, lambda g: g.contiguous(), lambda t: t.clone())
        for br in self.branches.values():
            for L in [br["enc"], br["corr"]] + br["tower"] + [br["pred"]]:
                add_sep(L)
        slots["adjust"] = (self.adjust, lambda g: g.reshape(1).contiguous(), lambda t: t.reshape(1).clone())
        slots["bias"] = (self.bias4, lambda g: g.reshape(4).contiguous(), lambda t: t.reshape(1, 4, 1, 1).clone())
        return slots

    def running_stats(self) -> Dict[str, torch.Tensor]:
        out = {}
        for br in self.branches.values():
            for L in [br["enc"], br["corr"]] + br["tower"]:
                out[L.bn_prefix + ".running_mean"] = L.running_mean
                out[L.bn_prefix + ".running_var"] = L.running_var
        return out
