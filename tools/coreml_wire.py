"""Minimal protobuf wire-format reader for CoreML ``NeuralNetwork`` specs.

Test/tooling infrastructure: decodes the three ``*.mlmodel`` files the reference ships
(``evaluate/MeasurePerformance/MeasurePerformance/models/FEAR-XS-NoEmbs.mlmodel``,
``evaluate/FEARDemo/FEARDemo/Tracker.mlmodel``, ``.../TrackerInit.mlmodel``) without
coremltools (not installed, no network).  Those files were produced by the reference's
``evaluate/coreml_convert.py:60-70`` + ``evaluate/coreml_utils.py:18-58`` and are the only
place the trained, BN-folded FEAR-XS weights exist in the repository (the Lightning
checkpoint is listed in ``.MISSING_LARGE_BLOBS``).

Field numbers follow Apple's public ``Model.proto`` / ``NeuralNetwork.proto``
(coremltools 5.1): Model.neuralNetwork=500, NeuralNetwork.layers=1, .preprocessing=2,
NeuralNetworkLayer.{name=1,input=2,output=3,convolution=100,activation=130,unary=220,
add=230,scale=245,concat=320,transpose=985,batchedMatmul=1045,reshapeStatic=1140}.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Tuple

import numpy as np


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """Yield (field_number, wire_type, value) for one message body."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt} at {pos}")
        yield fno, wt, val


def _packed_varints(b) -> List[int]:
    if isinstance(b, int):
        return [b]
    out = []
    pos = 0
    while pos < len(b):
        v, pos = _varint(b, pos)
        out.append(v)
    return out


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _weight_params(buf: bytes) -> np.ndarray:
    """WeightParams: floatValue=1 (packed f32) | float16Value=2 (bytes, LE f16)."""
    f32 = None
    f16 = None
    for fno, wt, val in fields(buf):
        if fno == 1:
            if wt == 2:
                f32 = np.frombuffer(val, dtype="<f4")
            else:
                f32 = np.frombuffer(val, dtype="<f4")
        elif fno == 2:
            f16 = np.frombuffer(val, dtype="<f2")
    if f16 is not None and f16.size:
        return f16.copy()
    if f32 is not None:
        return f32.copy()
    return np.zeros(0, dtype="<f2")


def _conv(buf: bytes) -> Dict:
    d = {"groups": 1, "kernel": [3, 3], "stride": [1, 1], "dilation": [1, 1], "pad": [0, 0, 0, 0],
         "has_bias": False, "weights": None, "bias": None}
    for fno, wt, val in fields(buf):
        if fno == 1:
            d["cout"] = val
        elif fno == 2:
            d["cin_per_group"] = val
        elif fno == 10:
            d["groups"] = val
        elif fno == 20:
            d["kernel"] = _packed_varints(val)
        elif fno == 30:
            d["stride"] = _packed_varints(val)
        elif fno == 40:
            d["dilation"] = _packed_varints(val)
        elif fno == 50:  # ValidPadding{ paddingAmounts=1: BorderAmounts{ borderAmounts=10: EdgeSizes{1,2}* } }
            edges = []
            for f2, _, v2 in fields(val):
                if f2 == 1:
                    for f3, _, v3 in fields(v2):
                        if f3 == 10:
                            s = e = 0
                            for f4, _, v4 in fields(v3):
                                if f4 == 1:
                                    s = v4
                                elif f4 == 2:
                                    e = v4
                            edges.append((s, e))
            if edges:
                (t, b), (l, r) = edges[0], edges[1]
                d["pad"] = [t, b, l, r]
        elif fno == 70:
            d["has_bias"] = bool(val)
        elif fno == 90:
            d["weights"] = _weight_params(val)
        elif fno == 91:
            d["bias"] = _weight_params(val)
    return d


_LAYER_KINDS = {100: "conv", 130: "activation", 220: "unary", 230: "add", 245: "scale",
                320: "concat", 985: "transpose", 1045: "batched_matmul", 1140: "reshape_static"}


def _layer(buf: bytes) -> Dict:
    lay = {"name": "", "inputs": [], "outputs": [], "kind": None}
    for fno, wt, val in fields(buf):
        if fno == 1:
            lay["name"] = val.decode()
        elif fno == 2:
            lay["inputs"].append(val.decode())
        elif fno == 3:
            lay["outputs"].append(val.decode())
        elif fno in _LAYER_KINDS:
            kind = _LAYER_KINDS[fno]
            lay["kind"] = kind
            if kind == "conv":
                lay.update(_conv(val))
            elif kind == "activation":
                sub = [f for f, _, _ in fields(val)]
                lay["act"] = "relu" if 10 in sub else f"unknown{sub}"
            elif kind == "unary":
                u = {"type": 0, "alpha": 1.0, "epsilon": 1e-6, "shift": 0.0, "scale": 1.0}
                for f2, w2, v2 in fields(val):
                    if f2 == 1:
                        u["type"] = v2
                    elif f2 == 2:
                        u["alpha"] = struct.unpack("<f", v2)[0]
                    elif f2 == 3:
                        u["epsilon"] = struct.unpack("<f", v2)[0]
                    elif f2 == 4:
                        u["shift"] = struct.unpack("<f", v2)[0]
                    elif f2 == 5:
                        u["scale"] = struct.unpack("<f", v2)[0]
                lay["unary"] = u
            elif kind == "scale":
                for f2, w2, v2 in fields(val):
                    if f2 == 1:
                        lay["shape_scale"] = _packed_varints(v2)
                    elif f2 == 2:
                        lay["scale"] = _weight_params(v2)
                    elif f2 == 3:
                        lay["has_bias"] = bool(v2)
                    elif f2 == 5:
                        lay["bias"] = _weight_params(v2)
            elif kind == "transpose":
                for f2, w2, v2 in fields(val):
                    if f2 == 1:
                        lay["axes"] = _packed_varints(v2)
            elif kind == "reshape_static":
                for f2, w2, v2 in fields(val):
                    if f2 == 1:
                        lay["shape"] = [_signed64(x) for x in _packed_varints(v2)]
            elif kind == "batched_matmul":
                bm = {"transpose_a": False, "transpose_b": False}
                for f2, w2, v2 in fields(val):
                    if f2 == 1:
                        bm["transpose_a"] = bool(v2)
                    elif f2 == 2:
                        bm["transpose_b"] = bool(v2)
                lay["matmul"] = bm
    return lay


def decode_mlmodel(path: str) -> Dict:
    """Return {"layers": [...], "preprocessing": {...}, "spec_version": int}."""
    with open(path, "rb") as fh:
        buf = fh.read()
    out = {"layers": [], "preprocessing": None, "spec_version": None}
    for fno, wt, val in fields(buf):
        if fno == 1:
            out["spec_version"] = val
        elif fno == 500:
            for f2, w2, v2 in fields(val):
                if f2 == 1:
                    out["layers"].append(_layer(v2))
                elif f2 == 2:
                    pp = {}
                    for f3, w3, v3 in fields(v2):
                        if f3 == 1:
                            pp["feature"] = v3.decode()
                        elif f3 == 10:
                            for f4, w4, v4 in fields(v3):
                                name = {10: "channel_scale", 20: "blue_bias", 21: "green_bias",
                                        22: "red_bias", 30: "gray_bias"}.get(f4)
                                if name:
                                    pp[name] = struct.unpack("<f", v4)[0]
                    out["preprocessing"] = pp
    return out
