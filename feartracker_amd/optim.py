"""`torch.optim.Adam` for the HIP training step: the reference's optimiser is Adam(lr = 1e-4) on every parameter
(model_training/train/base_lightning_model.py:63-64).

    net = FEARNetTrainHIP(state)
    opt = AdamHIP(net)                       # lr 1e-4, betas (0.9, 0.999), eps 1e-8, no weight decay: torch's defaults
    out = net.step(template, search, gt_reg, gt_cls, gt_weight)
    opt.step(net.allreduce_gradients(out["grads"]))      # (the all-reduce only with several ranks)

The update runs on the device, on the tensors the kernels read (`parameter_slots`: kernel layouts, so no re-layout of the
weights between steps); first / second moments are kept in the same layout.  One `fear_adam_step` launch per parameter tensor
(195 for the whole network, ~1.4 M floats): the learning-rate schedule of the reference (ReduceLROnPlateau) only changes `lr`.
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
        self.exp_avg = {k: torch.zeros_like(t) for k, (t, _, _) in self.slots.items()}
        self.exp_avg_sq = {k: torch.zeros_like(t) for k, (t, _, _) in self.slots.items()}
        self.steps = 0

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
            for name, (param, to_storage, _) in self.slots.items():
                g = to_storage(grads[name].to(dev, torch.float32))
                if g.shape != param.shape:
                    raise ValueError(f"{name}: gradient {tuple(grads[name].shape)} does not fit parameter storage {tuple(param.shape)}")
                rc = self.lib.fear_adam_step(_p(param), _p(g), _p(self.exp_avg[name]), _p(self.exp_avg_sq[name]), param.numel(),
                                             self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.steps, st)
                if rc != 0:
                    raise TrainError(f"fear_adam_step failed with status {rc} on {name}")
