"""`torch.optim.Adam` for the HIP training step: the reference's optimiser is Adam(lr = 1e-4) on every parameter
(model_training/train/base_lightning_model.py:63-64).

    net = FEARNetTrainHIP(state)
    opt = AdamHIP(net)                       # lr 1e-4, betas (0.9, 0.999), eps 1e-8, no weight decay: torch's defaults
    out = net.step(template, search, gt_reg, gt_cls, gt_weight)
    opt.step(net.allreduce_gradients(out["grads"]))      # (the all-reduce only with several ranks)

The update runs on the device, on the tensors the kernels read (kernel layouts, so no re-layout of the weights between steps);
first / second moments are kept in the same layout.  `FEARNetTrainHIP` keeps all 195 parameter tensors (1.37 M floats) in ONE
flat buffer (`param_flat`) and hands its gradients out as views of one buffer of the same layout (`GradDict.flat`): the whole
update is ONE `fear_adam_step` launch over (parameters, gradients, moments) — it was one launch and three small torch ops per
tensor, 1.6 ms of launches for 5 MB of data.  Gradients that arrive as a plain dict in the reference's layouts (the checker's
autograd in the tests) are first laid out into a staging buffer of that layout; a model without `param_flat` (the head alone,
`BoxTowerTrainHIP`) is updated tensor by tensor.  The learning-rate schedule of the reference (ReduceLROnPlateau) only changes
`lr`.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Tuple

import torch

from .train_head import TrainError, _p, load_train_library


class AdamHIP:
    def __init__(self, net, lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        self.lib = load_train_library()
        self.net = net
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.slots = net.parameter_slots()
        self.flat = getattr(net, "param_flat", None)
        if self.flat is not None:
            base = self.flat.data_ptr()
            self._off = {}
            for k, (t, _, _) in self.slots.items():
                off = (t.data_ptr() - base) // 4
                if not (0 <= off and off + t.numel() <= self.flat.numel() and t.is_contiguous()):
                    raise ValueError(f"{k}: parameter storage is not a slot of the network's flat buffer")
                self._off[k] = off
            self.exp_avg_flat = torch.zeros_like(self.flat)
            self.exp_avg_sq_flat = torch.zeros_like(self.flat)
            view = lambda buf, k, t: buf[self._off[k]: self._off[k] + t.numel()].view(t.shape)
            self.exp_avg = {k: view(self.exp_avg_flat, k, t) for k, (t, _, _) in self.slots.items()}
            self.exp_avg_sq = {k: view(self.exp_avg_sq_flat, k, t) for k, (t, _, _) in self.slots.items()}
            self._stage = None
        else:
            self.exp_avg = {k: torch.zeros_like(t) for k, (t, _, _) in self.slots.items()}
            self.exp_avg_sq = {k: torch.zeros_like(t) for k, (t, _, _) in self.slots.items()}
        self.steps = 0

    def _adam(self, param, grad, m, v, n, st, what):
        rc = self.lib.fear_adam_step(_p(param), _p(grad), _p(m), _p(v), n, self.lr, self.betas[0], self.betas[1], self.eps,
                                     self.weight_decay, self.steps, st)
        if rc != 0:
            raise TrainError(f"fear_adam_step failed with status {rc} on {what}")

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor]) -> None:
        """One Adam update from gradients in the reference's layouts ({parameter name: tensor}, as `net.step` returns them).
        Every parameter must have a gradient (the reference's graph leaves none unused)."""
        missing = [k for k in self.slots if k not in grads]
        if missing:
            raise KeyError(f"no gradient for {missing[:3]}{'...' if len(missing) > 3 else ''}")
        self.steps += 1
        dev = self.net.device
        with torch.cuda.device(dev):
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if self.flat is not None:
                # (GradDict.current_flat: None once an entry was rebound or a re-laid-out copy edited — the dict's values then count)
                gflat = grads.current_flat() if hasattr(grads, "current_flat") else getattr(grads, "flat", None)
                if gflat is None or gflat.numel() != self.flat.numel() or gflat.device != self.flat.device:
                    # gradients from somewhere else: lay them out like the parameters (padding between slots stays zero)
                    if self._stage is None:
                        self._stage = torch.zeros_like(self.flat)
                    gflat = self._stage
                    for name, (param, to_storage, _) in self.slots.items():
                        g = to_storage(grads[name].to(dev, torch.float32))
                        if g.shape != param.shape:
                            raise ValueError(f"{name}: gradient {tuple(grads[name].shape)} does not fit parameter storage {tuple(param.shape)}")
                        gflat[self._off[name]: self._off[name] + g.numel()].view(g.shape).copy_(g)
                self._adam(self.flat, gflat, self.exp_avg_flat, self.exp_avg_sq_flat, self.flat.numel(), st, "the flat parameter buffer")
                return
            for name, (param, to_storage, _) in self.slots.items():
                g = to_storage(grads[name].to(dev, torch.float32))
                if g.shape != param.shape:
                    raise ValueError(f"{name}: gradient {tuple(grads[name].shape)} does not fit parameter storage {tuple(param.shape)}")
                self._adam(param, g, self.exp_avg[name], self.exp_avg_sq[name], param.numel(), st, name)
