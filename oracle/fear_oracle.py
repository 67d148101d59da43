"""CPU oracle for the FEAR-XS per-frame inference path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  The product (`feartracker_amd/`) never does; it fails loudly when the HIP library
is missing.

What is restated here, and what pins it
---------------------------------------
* Trunk (`FEARNet.feature_extractor`, model_training/model/fear_net.py:58-61 over
  `Encoder.stages[:4]`, model/blocks.py:27-35): the architecture lives in the un-vendored
  third-party dependency `mobile_cv` (facebookresearch/mobile-vision @
  51804a6873ae1029257cf652179c960cceeecc75, requirements.txt:8), absent from
  /root/reference.  The restatement follows the published FBNet-V2 inverted-residual block
  (1x1 expand + ReLU -> depthwise kxk + ReLU -> 1x1 project (linear) [+ input]) with the block
  table recovered from the reference's own shipped trace of exactly that module
  (`FEAR-XS-NoEmbs.mlmodel`, written by evaluate/coreml_convert.py:60-70).  The reference has
  no tests/golden vectors for the trunk -> **trunk parity is pinned by that artefact only**:
  tests/golden/*.npz hold outputs of a literal layer-by-layer interpretation of the CoreML
  graph (tools/coreml_interp.py, no block structure assumed) which this oracle must match.
* Neck (`AdjustLayer`, blocks.py:75-88), head (`MatrixMobile` :91-105, `MobileCorrelation`
  :108-126, `BoxTower.forward` :174-194), decode (`FEARBoxCoder.decode`,
  model_training/dataset/box_coder.py:75-107, `make_grid` utils/utils.py:184-199): pinned
  against the reference's own Python classes imported in the build container
  (tools/make_golden.py) — fixtures in tests/golden/.

Arithmetic: fp32 throughout (torch CPU `F.conv2d`), weights = the fp16 values of the model
file upcast to fp32, exactly what the HIP engine uploads.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

K_STEM, K_IR, K_NECK, K_SEP = 0, 1, 2, 3
ROLE_CLS_ENCODE, ROLE_REG_ENCODE, ROLE_CLS_CORR, ROLE_REG_CORR = 1, 2, 3, 4
ROLE_BBOX_TOWER, ROLE_CLS_TOWER, ROLE_BBOX_PRED, ROLE_CLS_PRED = 5, 6, 7, 8
ACT_NONE, ACT_RELU, ACT_EXP = 0, 1, 2

# reference dict keys, model_training/utils/constants.py:1,3
TARGET_CLASSIFICATION_KEY = "TARGET_CLASSIFICATION_KEY"
TARGET_REGRESSION_LABEL_KEY = "TARGET_REGRESSION_LABEL_KEY"


def load_fearw(path: str) -> Dict:
    """Parse a `.fearw` file (layout: include/fearw_format.h) into fp32 torch tensors."""
    with open(path, "rb") as fh:
        buf = fh.read()
    magic, version, n_convs, n_blocks, dtype, payload_bytes = struct.unpack_from("<8s4IQ", buf, 0)
    if magic != b"FEARW1\0\0" or version != 1 or dtype not in (0, 1):
        raise ValueError(f"{path}: not a FEARW1 file with an fp16 / fp32 payload")
    ety = "<f4" if dtype == 1 else "<f2"
    off = 64
    convs = []
    entries = []
    for _ in range(n_convs):
        e = struct.unpack_from("<8I2Q24s", buf, off)
        off += 72
        entries.append(e)
    blocks = []
    for _ in range(n_blocks):
        kind, role, c0, c1, c2, residual, act, _r = struct.unpack_from("<2I3i3I", buf, off)
        off += 32
        blocks.append(dict(kind=kind, role=role, conv=[c0, c1, c2], residual=residual, act=act))
    payload = buf[off:off + payload_bytes]
    if len(payload) != payload_bytes:
        raise ValueError(f"{path}: truncated payload")
    for cout, cin_g, groups, k, stride, pad, relu, has_bias, w_off, b_off, name in entries:
        nw = cout * cin_g * k * k
        w = np.frombuffer(payload, dtype=ety, count=nw, offset=w_off).astype(np.float32)
        w = torch.from_numpy(w.reshape(cout, cin_g, k, k).copy())
        b = None
        if has_bias:
            b = torch.from_numpy(np.frombuffer(payload, dtype=ety, count=cout, offset=b_off).astype(np.float32).copy())
        convs.append(dict(w=w, b=b, groups=groups, k=k, stride=stride, pad=pad, relu=bool(relu),
                          name=name.rstrip(b"\0").decode()))
    return dict(convs=convs, blocks=blocks)


class OracleNet:
    """fp32 CPU restatement of `FEARNet.get_features/track` (fear_net.py:63-66, 90-96)."""

    def __init__(self, fearw_path: str):
        m = load_fearw(fearw_path)
        self.convs: List[Dict] = m["convs"]
        self.blocks: List[Dict] = m["blocks"]
        self.trunk = [b for b in self.blocks if b["kind"] in (K_STEM, K_IR)]
        self.neck = [b for b in self.blocks if b["kind"] == K_NECK]
        assert len(self.neck) == 1
        self.head = {}
        for b in self.blocks:
            if b["kind"] == K_SEP:
                self.head.setdefault(b["role"], []).append(b)

    # -- primitive: one folded conv (+ReLU if the graph has one right after it)
    def _conv(self, idx: int, x: torch.Tensor, relu: Optional[bool] = None) -> torch.Tensor:
        c = self.convs[idx]
        y = F.conv2d(x, c["w"], c["b"], stride=c["stride"], padding=c["pad"], groups=c["groups"])
        if c["relu"] if relu is None else relu:
            y = F.relu(y)
        return y

    def _ir(self, b: Dict, x: torch.Tensor) -> torch.Tensor:
        y = x
        if b["conv"][0] >= 0:
            y = self._conv(b["conv"][0], y)
        y = self._conv(b["conv"][1], y)
        y = self._conv(b["conv"][2], y)
        if b["residual"]:
            y = y + x
        return y

    def _sep(self, b: Dict, x: torch.Tensor) -> torch.Tensor:
        """SepConv (+folded BN) + activation, model/blocks.py:45-72."""
        y = self._conv(b["conv"][0], x)
        y = self._conv(b["conv"][1], y)
        if b["act"] == ACT_EXP:
            y = torch.exp(y)
        return y

    @torch.no_grad()
    def feature_extractor(self, x: torch.Tensor, taps: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        for b in self.trunk:
            if b["kind"] == K_STEM:
                x = self._conv(b["conv"][0], x)
            else:
                x = self._ir(b, x)
            if taps is not None:
                taps.append(x)
        return x

    @torch.no_grad()
    def get_features(self, crop: torch.Tensor) -> torch.Tensor:
        """fear_net.py:63-66: trunk then AdjustLayer (1x1 + BN, no activation)."""
        x = self.feature_extractor(crop.float())
        return self._conv(self.neck[0]["conv"][0], x)

    @torch.no_grad()
    def connector(self, template_features: torch.Tensor, search_features: torch.Tensor,
                  update: Optional[torch.Tensor] = None, return_all: bool = False):
        """`BoxTower.forward(search, kernel, update)`, model/blocks.py:174-194."""
        h = self.head
        x = search_features
        z_reg = template_features
        z_cls = template_features if update is None else update
        cls_x = self._sep(h[ROLE_CLS_ENCODE][0], x)
        reg_x = self._sep(h[ROLE_REG_ENCODE][0], x)

        def corr(z, xx, blk):
            # MobileCorrelation.forward, blocks.py:121-126
            b, c, hh, ww = xx.shape
            zf = z.reshape(z.size(0), z.size(1), -1)
            s = torch.matmul(zf.permute(0, 2, 1), xx.reshape(b, c, -1)).view(b, -1, hh, ww)
            s = torch.cat([xx, s], dim=1)
            return self._sep(blk, s)

        cls_dw = corr(z_cls, cls_x, h[ROLE_CLS_CORR][0])
        reg_dw = corr(z_reg, reg_x, h[ROLE_REG_CORR][0])
        x_reg = reg_dw
        for blk in h.get(ROLE_BBOX_TOWER, []):
            x_reg = self._sep(blk, x_reg)
        bbox = self._sep(h[ROLE_BBOX_PRED][0], x_reg)  # exp(adjust*conv+bias) folded, blocks.py:187-188
        c = cls_dw
        for blk in h.get(ROLE_CLS_TOWER, []):
            c = self._sep(blk, c)
        cls = self._sep(h[ROLE_CLS_PRED][0], c)  # 0.1 folded, blocks.py:192
        if return_all:
            return bbox, cls, cls_dw, x_reg
        return {TARGET_REGRESSION_LABEL_KEY: bbox, TARGET_CLASSIFICATION_KEY: cls}

    @torch.no_grad()
    def track(self, search: torch.Tensor, template_features: torch.Tensor,
              update: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """fear_net.py:90-96."""
        return self.connector(template_features, self.get_features(search), update)

    @torch.no_grad()
    def forward(self, x: Tuple[torch.Tensor, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """fear_net.py:83-88."""
        template, search = x
        return self.connector(self.get_features(template), self.get_features(search))


def normalize_u8(img_hwc_u8: np.ndarray) -> torch.Tensor:
    """`Tracker._preprocess_image` (base_tracker.py:97-103) with albumentations.Normalize
    (base_tracker.py:70-81): out = (px - 255*mean) * (1/(255*std)) in fp32, HWC -> 1xCxHxW.

    albumentations 1.0.0 `normalize`: mean*=max_pixel_value; std*=max_pixel_value;
    denominator = reciprocal(std, dtype=float32); img = img.astype(float32); img -= mean; img *= denominator
    (mean/std as float32 arrays).  cv2/albumentations are absent in this image: unpinned restatement.
    """
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32) * np.float32(255.0)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32) * np.float32(255.0)
    denom = np.reciprocal(std, dtype=np.float32)
    img = img_hwc_u8.astype(np.float32)
    img -= mean
    img *= denom
    return torch.from_numpy(np.transpose(img, (2, 0, 1))[None].copy())


def make_grid(score_size: int, total_stride: int, instance_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """utils/utils.py:184-199 — float64 grids of shape (1, S, S)."""
    x, y = np.meshgrid(
        np.arange(0, score_size) - np.floor(float(score_size // 2)),
        np.arange(0, score_size) - np.floor(float(score_size // 2)),
    )
    gx = x * total_stride + instance_size // 2
    gy = y * total_stride + instance_size // 2
    return gx[np.newaxis], gy[np.newaxis]


def decode(regression_map: np.ndarray, classification_map: np.ndarray, use_sigmoid: bool = True,
           score_size: int = 16, total_stride: int = 16, instance_size: int = 256):
    """`FEARBoxCoder.decode`, dataset/box_coder.py:75-107.

    regression_map (B,4,S,S) ltrb, classification_map (B,1,S,S).  Returns (bbox (B,4) float64
    xywh in crop pixels, list[(r, c)]).  First-maximum tie-breaking like torch.argmax.
    """
    gx, gy = make_grid(score_size, total_stride, instance_size)
    cm = classification_map.astype(np.float32)
    if use_sigmoid:
        cm = (1.0 / (1.0 + np.exp(-cm.astype(np.float64)))).astype(np.float32)
    cm = cm[:, 0]
    rm = regression_map
    x1 = gx - rm[:, 0]
    y1 = gy - rm[:, 1]
    x2 = gx + rm[:, 2]
    y2 = gy + rm[:, 3]
    boxes, coords = [], []
    for b in range(cm.shape[0]):
        idx = int(np.argmax(cm[b].reshape(-1)))
        r, c = idx // cm.shape[2], idx % cm.shape[2]
        o = [x1[b, r, c], y1[b, r, c], x2[b, r, c], y2[b, r, c]]
        boxes.append([o[0], o[1], o[2] - o[0], o[3] - o[1]])
        coords.append((r, c))
    return np.asarray(boxes, dtype=np.float64), coords
