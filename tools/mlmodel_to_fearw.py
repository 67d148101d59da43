#!/usr/bin/env python3
"""Convert a reference-shipped CoreML FEAR graph into the `.fearw` model file.

Tooling (runs in the build container where /root/reference is mounted; the GPU box only
ever sees the committed `.fearw` output).

Source of truth for the graph/weights: the fp16 `*.mlmodel` files the reference ships,
produced by `evaluate/coreml_convert.py:60-70` (`CoreMLTrackingWrapper.forward` ==
`FEARNet.track`, `model_training/model/fear_net.py:90-96`).  The converter

  1. decodes the protobuf (tools/coreml_wire.py),
  2. recovers the *block structure* of the BN-folded graph: stem, FBNet inverted-residual
     blocks (`mobile_cv` fbnet_c `stages[0:18]`, sliced by `model/blocks.py:27-35` and
     `fear_net.py:58-61`), the 1x1 neck (`blocks.py:75-88`) and the separable-conv head
     stages of `BoxTower` (`blocks.py:129-194`),
  3. writes the lossless fp16 payload plus conv/block tables (layout: include/fearw_format.h).

No arithmetic is changed: weights stay the exact fp16 values of the .mlmodel (OIHW,
`[Cout, Cin/groups, kH, kW]`), `adjust`/`bias` stay folded into bbox_pred.pointwise and
the 0.1 factor into cls_pred.pointwise exactly as coremltools folded them.
"""
from __future__ import annotations

import argparse
import os
import struct
import sys
from typing import Dict, List

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from coreml_wire import decode_mlmodel  # noqa: E402

MAGIC = b"FEARW1\0\0"
VERSION = 1

# block kinds
K_STEM, K_IR, K_NECK, K_SEP = 0, 1, 2, 3
# SEP roles (head stages, names follow model_training/model/blocks.py:143-168)
ROLE_NONE = 0
ROLE_CLS_ENCODE, ROLE_REG_ENCODE = 1, 2
ROLE_CLS_CORR, ROLE_REG_CORR = 3, 4
ROLE_BBOX_TOWER, ROLE_CLS_TOWER = 5, 6
ROLE_BBOX_PRED, ROLE_CLS_PRED = 7, 8
# final activation of a block's last conv
ACT_NONE, ACT_RELU, ACT_EXP = 0, 1, 2


def analyse(model: Dict) -> Dict:
    """Turn the flat CoreML layer list into conv + block tables."""
    layers = model["layers"]
    producer = {}
    consumers: Dict[str, List[int]] = {}
    for i, lay in enumerate(layers):
        for o in lay["outputs"]:
            producer[o] = i
        for t in lay["inputs"]:
            consumers.setdefault(t, []).append(i)

    convs = []  # dicts with conv params (+relu flag)
    conv_of_layer = {}

    def relu_after(tensor: str):
        """If the only consumer set of `tensor` contains a ReLU activation return its output name."""
        for ci in consumers.get(tensor, []):
            if layers[ci]["kind"] == "activation" and layers[ci].get("act") == "relu":
                return layers[ci]["outputs"][0]
        return None

    for i, lay in enumerate(layers):
        if lay["kind"] != "conv":
            continue
        k = lay["kernel"]
        assert k[0] == k[1] and lay["stride"][0] == lay["stride"][1] and lay["dilation"] == [1, 1]
        pad = lay["pad"]
        assert pad[0] == pad[1] == pad[2] == pad[3] == k[0] // 2, (lay["name"], pad)
        w = lay["weights"]
        assert w.dtype == np.dtype("<f2"), "expected fp16 weights (quantised mlmodel)"
        assert w.size == lay["cout"] * lay["cin_per_group"] * k[0] * k[1]
        b = lay["bias"] if lay["has_bias"] else None
        if b is not None:
            assert b.dtype == np.dtype("<f2") and b.size == lay["cout"]
        out_t = lay["outputs"][0]
        r = relu_after(out_t)
        conv_of_layer[i] = len(convs)
        convs.append(dict(name=lay["name"], cout=lay["cout"], cin_g=lay["cin_per_group"], groups=lay["groups"],
                          k=k[0], stride=lay["stride"][0], pad=pad[0], relu=int(r is not None),
                          w=w, b=b, in_t=lay["inputs"][0], out_t=out_t, post_t=r or out_t, layer=i))

    def is_dw(c):
        return c["groups"] == c["cout"] and c["cin_g"] == 1 and c["groups"] > 1

    def is_pw(c):
        return c["groups"] == 1 and c["k"] == 1

    def add_consumer(tensor):
        for ci in consumers.get(tensor, []):
            if layers[ci]["kind"] == "add" and layers[ci]["inputs"][0] == tensor:
                return ci
        return None

    blocks = []
    ci = 0
    n = len(convs)
    # ---- stem
    c0 = convs[0]
    assert c0["groups"] == 1 and c0["k"] == 3 and c0["stride"] == 2 and c0["relu"], "unexpected stem"
    blocks.append(dict(kind=K_STEM, role=0, conv=[0, -1, -1], residual=0, act=ACT_RELU))
    cur = c0["post_t"]
    ci = 1
    # ---- trunk: inverted residual blocks until the neck (a pw conv without relu whose input is `cur`
    #      and whose output feeds depthwise convs of the head)
    template_reshape = [l for l in layers if l["kind"] == "reshape_static" and l["inputs"] == ["template_features"]]
    has_head = len(template_reshape) == 1
    while ci < n:
        c = convs[ci]
        assert c["in_t"] == cur, (c["name"], c["in_t"], cur)
        if is_pw(c) and not c["relu"]:
            nxt = convs[ci + 1] if ci + 1 < n else None
            # neck: linear 1x1 that is the last conv or is followed by the two head dw convs reading it
            if nxt is None or (is_dw(nxt) and nxt["in_t"] == c["out_t"] and not nxt["relu"] and nxt["b"] is None):
                blocks.append(dict(kind=K_NECK, role=0, conv=[ci, -1, -1], residual=0, act=ACT_NONE))
                cur = c["out_t"]
                ci += 1
                break
        if is_pw(c):
            assert c["relu"], f"expand conv {c['name']} without ReLU"
            exp, dw, prj = ci, ci + 1, ci + 2
        else:
            exp, dw, prj = -1, ci, ci + 1
        cd, cp = convs[dw], convs[prj]
        assert is_dw(cd) and cd["relu"] and is_pw(cp) and not cp["relu"], (cd["name"], cp["name"])
        if exp >= 0:
            assert cd["in_t"] == convs[exp]["post_t"]
        assert cp["in_t"] == cd["post_t"]
        a = add_consumer(cp["out_t"])
        residual = 0
        out_t = cp["out_t"]
        if a is not None and layers[a]["inputs"][0] == cp["out_t"]:
            # the block's own residual add is add(project_out, block_input); an add that has the
            # project output as its *second* operand belongs to the following block.
            assert layers[a]["inputs"][1] == cur, (layers[a]["inputs"], cur)
            residual = 1
            out_t = layers[a]["outputs"][0]
        blocks.append(dict(kind=K_IR, role=0, conv=[exp, dw, prj], residual=residual, act=ACT_NONE))
        cur = out_t
        ci = prj + 1
    n_trunk_convs = ci

    if has_head:
        feat = cur  # neck output
        by_out = {c["post_t"]: idx for idx, c in enumerate(convs)}

        def sep_from(tensor_in, role, expect_relu):
            """Find dw conv reading `tensor_in` (not yet used), then its pw."""
            for idx in range(n_trunk_convs, n):
                c = convs[idx]
                if c["in_t"] == tensor_in and is_dw(c) and idx not in used:
                    p = convs[idx + 1]
                    assert is_pw(p) and p["in_t"] == c["out_t"]
                    used.add(idx)
                    used.add(idx + 1)
                    return idx, idx + 1
            raise AssertionError(f"no sep conv reading {tensor_in} for role {role}")

        used = set()
        # encode stages: two dw(no bias) reading the neck output, in graph order cls then reg
        # (BoxTower.forward, blocks.py:176-181: cls_encode first, then reg_encode).
        enc_a = sep_from(feat, ROLE_CLS_ENCODE, True)
        enc_b = sep_from(feat, ROLE_REG_ENCODE, True)
        # which one is cls: follow to output named 'cls'
        def follow(enc):
            """Walk encode -> concat -> corr sep -> towers -> pred; return list of (dw,pw) and final tensor."""
            chain = []
            t = convs[enc[1]]["post_t"]
            # concat consumer
            cc = [i for i in consumers[t] if layers[i]["kind"] == "concat"]
            assert len(cc) == 1
            cat = layers[cc[0]]
            assert cat["inputs"][0] == t, "concat order must be [x, corr] (blocks.py:124)"
            # verify correlation path: reshape(x) -> matmul(transpose(z), .) -> reshape
            s_t = cat["inputs"][1]
            rs = layers[producer[s_t]]
            assert rs["kind"] == "reshape_static"
            mm = layers[producer[rs["inputs"][0]]]
            assert mm["kind"] == "batched_matmul" and not mm["matmul"]["transpose_a"] and not mm["matmul"]["transpose_b"]
            tr = layers[producer[mm["inputs"][0]]]
            assert tr["kind"] == "transpose" and tr["axes"] == [0, 2, 1]
            assert layers[producer[tr["inputs"][0]]]["inputs"] == ["template_features"]
            xr = layers[producer[mm["inputs"][1]]]
            assert xr["kind"] == "reshape_static" and xr["inputs"] == [t]
            t = cat["outputs"][0]
            while True:
                try:
                    d, p = sep_from(t, 0, True)
                except AssertionError:
                    break
                chain.append((d, p))
                t = convs[p]["post_t"]
            return chain, t

        chain_a, end_a = follow(enc_a)
        chain_b, end_b = follow(enc_b)

        def final_name(t):
            # bbox goes through unary exp
            cons = consumers.get(t, [])
            for i in cons:
                if layers[i]["kind"] == "unary":
                    assert layers[i]["unary"]["type"] == 4, "expected EXP"
                    return layers[i]["outputs"][0], ACT_EXP
            return t, ACT_NONE

        name_a, act_a = final_name(end_a)
        name_b, act_b = final_name(end_b)
        assert {name_a, name_b} == {"cls", "bbox"}, (name_a, name_b)
        if name_a == "cls":
            cls_enc, cls_chain, reg_enc, reg_chain = enc_a, chain_a, enc_b, chain_b
            assert act_b == ACT_EXP and act_a == ACT_NONE
        else:
            cls_enc, cls_chain, reg_enc, reg_chain = enc_b, chain_b, enc_a, chain_a
            assert act_a == ACT_EXP and act_b == ACT_NONE

        def emit(role, pair, act):
            d, p = pair
            assert (convs[p]["relu"] == 1) == (act == ACT_RELU), (convs[p]["name"], act)
            blocks.append(dict(kind=K_SEP, role=role, conv=[d, p, -1], residual=0, act=act))

        emit(ROLE_CLS_ENCODE, cls_enc, ACT_RELU)
        emit(ROLE_REG_ENCODE, reg_enc, ACT_RELU)
        emit(ROLE_CLS_CORR, cls_chain[0], ACT_RELU)
        emit(ROLE_REG_CORR, reg_chain[0], ACT_RELU)
        for pair in reg_chain[1:-1]:
            emit(ROLE_BBOX_TOWER, pair, ACT_RELU)
        for pair in cls_chain[1:-1]:
            emit(ROLE_CLS_TOWER, pair, ACT_RELU)
        emit(ROLE_BBOX_PRED, reg_chain[-1], ACT_EXP)
        emit(ROLE_CLS_PRED, cls_chain[-1], ACT_NONE)
        assert len(used) == n - n_trunk_convs, "unclaimed head convs"
    else:
        assert ci == n, "trunk-only model has trailing convs"

    return dict(convs=convs, blocks=blocks, preprocessing=model["preprocessing"])


def write_fearw(path: str, ana: Dict) -> None:
    convs, blocks = ana["convs"], ana["blocks"]
    payload = bytearray()
    table = bytearray()
    for c in convs:
        w_off = len(payload)
        payload += c["w"].astype("<f2").tobytes()
        if c["b"] is not None:
            b_off = len(payload)
            payload += c["b"].astype("<f2").tobytes()
            has_b = 1
        else:
            b_off, has_b = 0, 0
        while len(payload) % 16:
            payload += b"\0"
        name = c["name"].encode()[:23]
        table += struct.pack("<8I2Q24s", c["cout"], c["cin_g"], c["groups"], c["k"], c["stride"], c["pad"],
                             c["relu"], has_b, w_off, b_off, name)
    btab = bytearray()
    for b in blocks:
        btab += struct.pack("<2I3i3I", b["kind"], b["role"], b["conv"][0], b["conv"][1], b["conv"][2],
                            b["residual"], b["act"], 0)
    header = struct.pack("<8s4IQ", MAGIC, VERSION, len(convs), len(blocks), 0, len(payload))
    header += b"\0" * (64 - len(header))
    with open(path, "wb") as fh:
        fh.write(header)
        fh.write(table)
        fh.write(btab)
        fh.write(payload)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("mlmodel")
    ap.add_argument("out")
    args = ap.parse_args()
    model = decode_mlmodel(args.mlmodel)
    ana = analyse(model)
    write_fearw(args.out, ana)
    nparams = sum(c["w"].size + (0 if c["b"] is None else c["b"].size) for c in ana["convs"])
    print(f"{args.out}: {len(ana['convs'])} convs, {len(ana['blocks'])} blocks, {nparams} params")
    for b in ana["blocks"]:
        names = [ana["convs"][i]["name"] if i >= 0 else "-" for i in b["conv"]]
        print(b["kind"], b["role"], names, "res" if b["residual"] else "", b["act"])


if __name__ == "__main__":
    main()
