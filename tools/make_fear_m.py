#!/usr/bin/env python3
"""Write `feartracker_amd/weights/fear_m_synth.fearw`: the synthetic "FEAR-M" of BASELINE.json configs[3].

The reference defines no FEAR-M (model/blocks.py:22-25 hard-codes `fbnet_c`; SURVEY.md §7 "Hard parts"), so this is a
perf / numerics configuration with NO reference definition and NO trained weights — stated wherever its numbers appear:

  architecture  "deeper FBNet": the FEAR-XS block table (read from fear_xs_noembs.fearw) with every residual
                inverted-residual block repeated twice — 1+2+3+3+3 residual blocks become 2+4+6+6+6, the stride-2 /
                channel-changing blocks, the neck and the head stay as they are.  28 IR blocks, 660 M MAC per 256x256
                search crop (FEAR-XS: 16 blocks, 461 M).
  weights       seeded random init (numpy RandomState(1234)), then one layer-sequential variance normalisation pass on
                a seeded batch (LSUV-style, plain torch conv2d here — no repo or reference code involved): every conv's
                weights are rescaled so that its output has unit standard deviation (0.3 of the block input's for the
                residual projections, 0.5 for bbox_pred before exp), which keeps 40 layers of random weights inside
                fp16/bf16-friendly ranges; rounded to fp16 like every .fearw payload.

usage: python tools/make_fear_m.py [out.fearw]
"""
from __future__ import annotations

import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from mlmodel_to_fearw import write_fearw  # noqa: E402

K_STEM, K_IR, K_NECK, K_SEP = 0, 1, 2, 3


def read_structure(path):
    """Conv shapes + block table of a .fearw file (include/fearw_format.h); weights are not needed."""
    buf = open(path, "rb").read()
    magic, version, n_convs, n_blocks, dtype, payload_bytes = struct.unpack_from("<8s4IQ", buf, 0)
    assert magic == b"FEARW1\0\0" and version == 1
    off, convs, blocks = 64, [], []
    for _ in range(n_convs):
        cout, cin_g, groups, k, stride, pad, relu, has_bias, w_off, b_off, name = struct.unpack_from("<8I2Q24s", buf, off)
        off += 72
        convs.append(dict(cout=cout, cin_g=cin_g, groups=groups, k=k, stride=stride, pad=pad, relu=relu, has_bias=has_bias,
                          name=name.split(b"\0")[0].decode()))
    for _ in range(n_blocks):
        kind, role, c0, c1, c2, residual, act, _r = struct.unpack_from("<2I3i3I", buf, off)
        off += 32
        blocks.append(dict(kind=kind, role=role, conv=[c0, c1, c2], residual=residual, act=act))
    return convs, blocks


def build(seed: int = 1234):
    import torch
    import torch.nn.functional as F
    xs_convs, xs_blocks = read_structure(os.path.join(REPO, "feartracker_amd", "weights", "fear_xs_noembs.fearw"))
    rng = np.random.RandomState(seed)
    convs, blocks = [], []

    def new_conv(shape, tag=""):
        fan_in = shape["cin_g"] * shape["k"] * shape["k"]
        w = rng.standard_normal(shape["cout"] * fan_in) * np.sqrt(1.0 / fan_in)
        b = rng.standard_normal(shape["cout"]) * 0.05 if shape["has_bias"] else None
        convs.append(dict(shape, w=w.astype(np.float32), b=None if b is None else b.astype(np.float32),
                          name=(tag + shape["name"])[:23]))
        return len(convs) - 1

    for bi, b in enumerate(xs_blocks):
        for rep in range(2 if (b["kind"] == K_IR and b["residual"]) else 1):
            ids = [new_conv(xs_convs[ci], tag=f"m{bi}r{rep}_") if ci >= 0 else -1 for ci in b["conv"]]
            blocks.append(dict(kind=b["kind"], role=b["role"], conv=ids, residual=b["residual"], act=b["act"]))

    # ---- variance normalisation: run the graph once, rescaling each conv to the target output std as it is reached
    g = torch.Generator().manual_seed(seed)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1) * 255.0
    inv = 1.0 / (torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1) * 255.0)
    search = (torch.randint(0, 256, (2, 3, 256, 256), generator=g).float() - mean) * inv
    tmpl = (torch.randint(0, 256, (2, 3, 128, 128), generator=g).float() - mean) * inv

    def run(ci, x, target=None, act_relu=None):
        c = convs[ci]
        w = torch.from_numpy(c["w"]).view(c["cout"], c["cin_g"], c["k"], c["k"])
        b = None if c["b"] is None else torch.from_numpy(c["b"])
        y = F.conv2d(x, w, None, stride=c["stride"], padding=c["pad"], groups=c["groups"])
        if target is not None:
            sc = float(target / y.std().clamp_min(1e-12))
            c["w"] = (c["w"] * sc).astype(np.float16).astype(np.float32)       # the payload is fp16: normalise what is stored
            w = torch.from_numpy(c["w"]).view(c["cout"], c["cin_g"], c["k"], c["k"])
            y = F.conv2d(x, w, None, stride=c["stride"], padding=c["pad"], groups=c["groups"])
        if b is not None:
            c["b"] = c["b"].astype(np.float16).astype(np.float32)
            y = y + torch.from_numpy(c["b"]).view(1, -1, 1, 1)
        return F.relu(y) if (c["relu"] if act_relu is None else act_relu) else y

    def trunk(x, tune):
        feats = None
        for b in blocks:
            if b["kind"] == K_STEM:
                x = run(b["conv"][0], x, 1.0 if tune else None)
            elif b["kind"] == K_IR:
                y = x
                if b["conv"][0] >= 0:
                    y = run(b["conv"][0], y, 1.0 if tune else None)
                y = run(b["conv"][1], y, 1.0 if tune else None)
                y = run(b["conv"][2], y, (0.3 * float(x.std()) if b["residual"] else 1.0) if tune else None)
                x = x + y if b["residual"] else y
            elif b["kind"] == K_NECK:
                feats = run(b["conv"][0], x, 1.0 if tune else None)
        return feats

    with torch.no_grad():
        xs = trunk(search, True)
        zs = trunk(tmpl, False)
        role = {}
        towers = {5: [], 6: []}
        for b in blocks:
            if b["kind"] == K_SEP:
                (towers[b["role"]].append(b) if b["role"] in towers else role.__setitem__(b["role"], b))

        def sep(b, x, target, relu):
            d = run(b["conv"][0], x, 1.0, act_relu=False)
            return run(b["conv"][1], d, target, act_relu=relu)

        for enc_r, corr_r, tower_r, pred_r in ((1, 3, 6, 8), (2, 4, 5, 7)):
            x = sep(role[enc_r], xs, 1.0, True)
            n = x.shape[0]
            corr = torch.matmul(zs.reshape(n, 256, 64).transpose(1, 2), x.reshape(n, 256, 256)).reshape(n, 64, 16, 16)
            x = sep(role[corr_r], torch.cat([x, corr], dim=1), 1.0, True)
            for tb in towers[tower_r]:
                x = sep(tb, x, 1.0, True)
            pred = sep(role[pred_r], x, 0.5 if pred_r == 7 else 1.0, False)
            if pred_r == 7:                       # ltrb distances of a few tens of pixels after exp
                pc = convs[role[7]["conv"][1]]
                pc["b"] = (pc["b"] + 3.0).astype(np.float16).astype(np.float32)
    return dict(convs=convs, blocks=blocks)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "feartracker_amd", "weights", "fear_m_synth.fearw")
    ana = build()
    write_fearw(out, ana)
    macs = 0
    n_ir = sum(1 for b in ana["blocks"] if b["kind"] == K_IR)
    nparams = sum(c["w"].size + (0 if c["b"] is None else c["b"].size) for c in ana["convs"])
    print(f"{out}: {len(ana['convs'])} convs, {len(ana['blocks'])} blocks ({n_ir} IR), {nparams} params, "
          f"{os.path.getsize(out)} bytes")


if __name__ == "__main__":
    main()
