"""Training state -> `.fearw` inference weights: the deployment step between `FEARNetTrainHIP` and `FEARNetHIP`.

The reference goes from a Lightning checkpoint to its deployed models by tracing `FEARNet.track` in eval mode and letting the
converter fold every BatchNorm into the convolution in front of it (evaluate/coreml_convert.py:60-70; the shipped
`FEAR-XS-NoEmbs.mlmodel` is the result, fp16).  `fold_training_state` does that folding explicitly on a state dict with the
parameter names of `train_net.py` (trunk: `stem.{conv,bn}`, `trunk.<i>.{pw,dw,pwl}.{conv,bn}`, `neck.downsample.{0,1}`) and of
the reference's own `BoxTower` (head: `connect_model.*`, model/blocks.py:129-194), and `write_fearw` stores the result in the
layout of include/fearw_format.h — the same block table the `.mlmodel` importer produces:

* conv + BatchNorm (running statistics):  w' = w * g,  b' = (b - mean) * g + beta,  g = gamma / sqrt(var + eps);
* SepConv + BatchNorm (head): the BatchNorm folds into the pointwise conv, the depthwise conv keeps its own bias;
* `bbox_pred`:  exp(adjust * conv(x) + bias)  ->  pointwise weights and bias scaled by `adjust`, `bias` added, act = exp
  (blocks.py:185-187);  `cls_pred`:  0.1 * conv(x)  ->  pointwise scaled by 0.1 (blocks.py:190).

payload = "fp16" rounds to half precision like the reference's deployed models (and keeps the weights exact for the engine's
fp16-split arithmetic mode); "fp32" stores the folded weights as they are (FEARW_PAYLOAD_F32).
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

from .train_net import TRUNK_BLOCKS

K_STEM, K_IR, K_NECK, K_SEP = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_EXP = 0, 1, 2
ROLE = dict(cls_encode=1, reg_encode=2, cls_corr=3, reg_corr=4, bbox_tower=5, cls_tower=6, bbox_pred=7, cls_pred=8)


def _np(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v, dtype=np.float64)


def fold_training_state(state: Dict[str, "np.ndarray"], eps: float = 1e-5) -> Tuple[List[dict], List[dict]]:
    """(convs, blocks) of include/fearw_format.h with every BatchNorm folded (float64 arithmetic, fp32 results)."""
    sd = {k: _np(v) for k, v in state.items()}
    convs: List[dict] = []

    def bn_scale(prefix):
        g = sd[prefix + ".weight"] / np.sqrt(sd[prefix + ".running_var"] + eps)
        return g, sd[prefix + ".bias"] - sd[prefix + ".running_mean"] * g

    def add(name, w, b, *, k, stride=1, groups=1, relu=0):
        cout, cin_g = w.shape[0], w.shape[1]
        convs.append(dict(cout=cout, cin_g=cin_g, groups=groups, k=k, stride=stride, pad=k // 2, relu=relu, name=name[:23],
                          w=np.ascontiguousarray(w, dtype=np.float32).reshape(-1),
                          b=None if b is None else np.ascontiguousarray(b, dtype=np.float32)))
        return len(convs) - 1

    def conv_bn(conv_key, bn_prefix, name, *, k, stride=1, depthwise=False, relu=0):
        w = sd[conv_key]
        g, shift = bn_scale(bn_prefix)
        return add(name, w * g.reshape(-1, 1, 1, 1), shift, k=k, stride=stride, groups=w.shape[0] if depthwise else 1, relu=relu)

    blocks: List[dict] = []
    blocks.append(dict(kind=K_STEM, role=0, conv=[conv_bn("stem.conv.weight", "stem.bn", "stem", k=3, stride=2, relu=1), -1, -1],
                       residual=0, act=ACT_RELU))
    for i, (cin, cexp, cout, k, stride, expand, residual) in enumerate(TRUNK_BLOCKS):
        p = f"trunk.{i}"
        ce = conv_bn(f"{p}.pw.conv.weight", f"{p}.pw.bn", f"t{i}_pw", k=1, relu=1) if expand else -1
        cd = conv_bn(f"{p}.dw.conv.weight", f"{p}.dw.bn", f"t{i}_dw", k=k, stride=stride, depthwise=True, relu=1)
        cp = conv_bn(f"{p}.pwl.conv.weight", f"{p}.pwl.bn", f"t{i}_pwl", k=1)
        blocks.append(dict(kind=K_IR, role=0, conv=[ce, cd, cp], residual=int(residual), act=ACT_NONE))
    blocks.append(dict(kind=K_NECK, role=0, conv=[conv_bn("neck.downsample.0.weight", "neck.downsample.1", "neck", k=1), -1, -1],
                       residual=0, act=ACT_NONE))

    def sep(prefix, name, role, *, bn=None, act=ACT_RELU, scale=None, shift=None):
        """SepConv at `prefix` (+ BatchNorm `bn`) (+ the affine scale / shift of the prediction heads) -> one FEARW_SEP block."""
        dwk, pwk = prefix + ".depthwise", prefix + ".pointwise"
        dw_w = sd[dwk + ".weight"]
        cd = add(name + "_dw", dw_w, sd.get(dwk + ".bias"), k=dw_w.shape[-1], groups=dw_w.shape[0])
        w = sd[pwk + ".weight"]
        b = sd.get(pwk + ".bias")
        b = np.zeros(w.shape[0]) if b is None else b
        if bn is not None:
            g, sh = bn_scale(bn)
            w, b = w * g.reshape(-1, 1, 1, 1), b * g + sh
        if scale is not None:
            w, b = w * scale, b * scale
        if shift is not None:
            b = b + shift
        cp = add(name + "_pw", w, b, k=1, relu=int(act == ACT_RELU))
        blocks.append(dict(kind=K_SEP, role=ROLE[role], conv=[cd, cp, -1], residual=0, act=act))

    h = "connect_model."
    sep(h + "cls_encode.matrix11_s.0", "cls_enc", "cls_encode", bn=h + "cls_encode.matrix11_s.1")
    sep(h + "reg_encode.matrix11_s.0", "reg_enc", "reg_encode", bn=h + "reg_encode.matrix11_s.1")
    sep(h + "cls_dw.enc.0", "cls_corr", "cls_corr", bn=h + "cls_dw.enc.1")
    sep(h + "reg_dw.enc.0", "reg_corr", "reg_corr", bn=h + "reg_dw.enc.1")
    towernum = len([k for k in sd if k.startswith(h + "bbox_tower.") and k.endswith(".depthwise.weight")])
    for t in range(towernum):
        sep(h + f"bbox_tower.{3 * t}", f"bbox_t{t}", "bbox_tower", bn=h + f"bbox_tower.{3 * t + 1}")
    for t in range(towernum):
        sep(h + f"cls_tower.{3 * t}", f"cls_t{t}", "cls_tower", bn=h + f"cls_tower.{3 * t + 1}")
    sep(h + "bbox_pred", "bbox_pred", "bbox_pred", act=ACT_EXP, scale=float(sd[h + "adjust"].reshape(-1)[0]),
        shift=sd[h + "bias"].reshape(-1))
    sep(h + "cls_pred", "cls_pred", "cls_pred", act=ACT_NONE, scale=0.1)
    return convs, blocks


def write_fearw(path: str, convs: List[dict], blocks: List[dict], payload: str = "fp16") -> None:
    """Write (convs, blocks) as a FEARW1 file (include/fearw_format.h); payload "fp16" or "fp32"."""
    if payload not in ("fp16", "fp32"):
        raise ValueError("payload must be 'fp16' or 'fp32'")
    ety, code = ("<f2", 0) if payload == "fp16" else ("<f4", 1)
    data, table = bytearray(), bytearray()
    for c in convs:
        w_off = len(data)
        data += np.asarray(c["w"], dtype=np.float32).astype(ety).tobytes()
        b_off, has_b = 0, 0
        if c["b"] is not None:
            b_off, has_b = len(data), 1
            data += np.asarray(c["b"], dtype=np.float32).astype(ety).tobytes()
        data += b"\0" * (-len(data) % 16)
        table += struct.pack("<8I2Q24s", c["cout"], c["cin_g"], c["groups"], c["k"], c["stride"], c["pad"], c["relu"], has_b,
                             w_off, b_off, c["name"].encode()[:23])
    btab = bytearray()
    for b in blocks:
        btab += struct.pack("<2I3i3I", b["kind"], b["role"], b["conv"][0], b["conv"][1], b["conv"][2], b["residual"], b["act"], 0)
    header = struct.pack("<8s4IQ", b"FEARW1\0\0", 1, len(convs), len(blocks), code, len(data))
    with open(path, "wb") as fh:
        fh.write(header + b"\0" * (64 - len(header)))
        fh.write(table)
        fh.write(btab)
        fh.write(data)


def export_training_state(state: Dict[str, "np.ndarray"], path: str, payload: str = "fp16", eps: float = 1e-5) -> None:
    """Fold the BatchNorms of a training state dict and write the `.fearw` file `FEARNetHIP(path)` loads."""
    convs, blocks = fold_training_state(state, eps=eps)
    write_fearw(path, convs, blocks, payload=payload)
