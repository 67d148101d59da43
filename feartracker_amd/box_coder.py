"""ltrb <-> xywh box coding on the 16x16 score grid.

Behavioural mirror of `FEARBoxCoder` (model_training/dataset/box_coder.py:53-107):
`encode` builds regression/classification targets, `decode` picks the arg-max cell of the
classification map and converts that cell's ltrb prediction to an xywh box in search-crop
pixels.  Grids are float64 like the reference's (`make_grid`, utils/utils.py:184-199), so
decoded boxes are float64.

Unlike the reference (a Python loop with one `.item()` device sync per sample,
box_coder.py:101-106) `decode` is vectorised: one arg-max + gather for the whole batch; when
the maps live on the GPU the HIP `fear_decode` kernel (include/fear_hip.h) does it on-device.
"""
from __future__ import annotations

from collections import namedtuple
from typing import Any, Dict, Union

import numpy as np
import torch

from .geometry import make_grid

TrackerEncodeResult = namedtuple("TrackerEncodeResult", ["regression_map", "classification_label"])
TrackerDecodeResult = namedtuple("TrackerDecodeResult", ["bbox", "pred_coords"])


class FEARBoxCoder:
    def __init__(self, tracker_config: Dict[str, Any]) -> None:
        self.tracker_config = tracker_config
        gx, gy = make_grid(tracker_config["score_size"], tracker_config["total_stride"],
                           tracker_config["instance_size"])
        self.grid_x = torch.from_numpy(gx)
        self.grid_y = torch.from_numpy(gy)

    def to_device(self, device: Union[str, int, torch.device]) -> "FEARBoxCoder":
        self.grid_x = self.grid_x.to(device)
        self.grid_y = self.grid_y.to(device)
        return self

    @torch.no_grad()
    def encode(self, bboxes: torch.Tensor) -> TrackerEncodeResult:
        """bboxes (B,4) xywh -> ltrb distances (B,4,S,S) fp32 and positive-cell mask (B,1,S,S)."""
        b = bboxes[:, :, None, None]
        x0, y0 = b[:, 0], b[:, 1]
        x1, y1 = x0 + b[:, 2], y0 + b[:, 3]
        ltrb = torch.stack((self.grid_x - x0, self.grid_y - y0, x1 - self.grid_x, y1 - self.grid_y), dim=1).float()
        inside = (ltrb.min(dim=1, keepdim=True).values > 0).float()
        return TrackerEncodeResult(regression_map=ltrb, classification_label=inside)

    @torch.no_grad()
    def decode(self, regression_map: torch.Tensor, classification_map: torch.Tensor,
               use_sigmoid: bool = True) -> TrackerDecodeResult:
        """regression_map (B,4,S,S) ltrb, classification_map (B,1,S,S) -> xywh (B,4), [(r,c)]."""
        score = classification_map
        if use_sigmoid:
            score = score.float().sigmoid()
        score = score[:, 0]
        n, s_h, s_w = score.shape
        flat = torch.argmax(score.reshape(n, -1), dim=1)            # first maximum, like the reference
        rows = torch.div(flat, s_w, rounding_mode="floor")
        cols = flat - rows * s_w
        ar = torch.arange(n, device=regression_map.device)
        gx = self.grid_x[0][rows, cols]
        gy = self.grid_y[0][rows, cols]
        l, t, r, b = (regression_map[ar, k, rows, cols] for k in range(4))
        x0, y0, x1, y1 = gx - l, gy - t, gx + r, gy + b
        boxes = torch.stack([x0, y0, x1 - x0, y1 - y0], dim=1)
        coords = list(zip(rows.tolist(), cols.tolist()))
        return TrackerDecodeResult(bbox=boxes, pred_coords=coords)
