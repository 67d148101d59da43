"""Literal layer-by-layer execution of a decoded CoreML NeuralNetwork graph on CPU torch.

Tooling for golden-vector generation only (needs the reference's .mlmodel files, so it runs
only in the build container).  It assumes NO block structure: every layer of the flat list
is executed as written (conv / ReLU / add / concat / reshape / transpose / batched matmul /
exp / per-channel scale), so it is an independent check of the block table recovered by
tools/mlmodel_to_fearw.py and of oracle/fear_oracle.py.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F


@torch.no_grad()
def run_graph(model: Dict, inputs: Dict[str, torch.Tensor], apply_scale_layer: bool = False,
              keep: Optional[set] = None, stop_after: Optional[str] = None) -> Dict[str, torch.Tensor]:
    """inputs: {"image": NCHW fp32 *already normalised* unless apply_scale_layer, "template_features": ...}.

    With apply_scale_layer=False the leading `scale_layer` (evaluate/coreml_utils.py:108-134) is treated as
    identity so the graph computes exactly `FEARNet.track(normalised_search, template_features)`.
    """
    env = dict(inputs)
    for lay in model["layers"]:
        k = lay["kind"]
        xs = [env[n] for n in lay["inputs"]]
        if k == "scale":
            if apply_scale_layer:
                s = torch.from_numpy(lay["scale"].astype(np.float32)).view(1, -1, 1, 1)
                y = xs[0] * s
            else:
                y = xs[0]
        elif k == "conv":
            kk = lay["kernel"][0]
            w = torch.from_numpy(lay["weights"].astype(np.float32)).view(lay["cout"], lay["cin_per_group"], kk, kk)
            b = torch.from_numpy(lay["bias"].astype(np.float32)) if lay["has_bias"] else None
            y = F.conv2d(xs[0], w, b, stride=lay["stride"][0], padding=lay["pad"][0], groups=lay["groups"])
        elif k == "activation":
            assert lay["act"] == "relu"
            y = F.relu(xs[0])
        elif k == "add":
            y = xs[0] + xs[1]
        elif k == "concat":
            y = torch.cat(xs, dim=1)
        elif k == "reshape_static":
            shape = list(lay["shape"])
            shape[0] = xs[0].shape[0]  # batch dim is 1 in the trace; generalise
            y = xs[0].reshape(shape)
        elif k == "transpose":
            y = xs[0].permute(*lay["axes"])
        elif k == "batched_matmul":
            y = torch.matmul(xs[0], xs[1])
        elif k == "unary":
            assert lay["unary"]["type"] == 4
            y = torch.exp(xs[0])
        else:
            raise NotImplementedError(k)
        env[lay["outputs"][0]] = y
        if stop_after is not None and lay["outputs"][0] == stop_after:
            break
    if keep is not None:
        return {n: env[n] for n in keep}
    return env
