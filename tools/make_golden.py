#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference).  The reference has no tests and
no golden vectors of its own, so every fixture is produced here by importing
the reference's Python modules — with import-time stubs for third-party packages that are not
installed (none of the stubbed symbols executes on the paths exercised) — and, for the trunk
that lives in the absent `mobile_cv` package, by literally interpreting the CoreML trace the
reference ships.  Only arrays are written; no reference source or bytecode enters the repo.

Fixtures
  grid_window.npz      make_grid(16,16,256), Hann window            (utils.py:184-199, base_tracker.py:58-67)
  geometry.npz         extend_bbox / ensure_bbox_boundaries / clamp_bbox on seeded boxes
  box_coder.npz        FEARBoxCoder.encode / decode on seeded maps   (box_coder.py:58-107)
  postprocess.npz      FEARTracker._postprocess smooth off/on, _rescale_bbox, _smooth_size
  head_modules.npz     reference AdjustLayer + BoxTower loaded with the CoreML weights
  track_maps.npz       whole-net track() on 8 seeded uint8 crops (literal CoreML graph): maps, argmax, boxes
  trunk_taps.npz       per-block trunk activations for one crop (kernel bring-up)
  clip_synth.npz       reference FEARTracker.initialize/update loop on a deterministic synthetic clip
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

_MISSING = {"mobile_cv", "cv2", "albumentations", "got10k", "coloredlogs", "hydra", "omegaconf",
            "pytorch_toolbelt", "pytorch_lightning", "torchmetrics", "torchvision", "fire", "imageio",
            "thop", "coremltools"}


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        return _Dummy()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if self.__name__ == "coloredlogs" and name == "DEFAULT_FIELD_STYLES":
            return {}
        return _Dummy


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _MISSING:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def import_reference():
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)
    from model_training.model import blocks as ref_blocks
    from model_training.dataset import box_coder as ref_box_coder
    from model_training.tracker import fear_tracker as ref_fear_tracker
    from model_training.utils import utils as ref_utils
    return ref_blocks, ref_box_coder, ref_fear_tracker, ref_utils


TRACKING_CONFIG = dict(penalty_k=0.062, window_influence=0.38, lr=0.765, windowing="cosine", total_stride=16,
                       score_size=16, ratio=0.94, stride=2, bbox_ratio=0.5, template_bbox_offset=0.2,
                       search_context=2, instance_size=256, template_size=128)


def synth_clip(n_frames=24, h=192, w=320, seed=7):
    """Deterministic RGB clip: textured background + a moving, slowly growing textured ellipse."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    bg = (40 + 30 * np.sin(xx / 23.0)[..., None] + 25 * np.cos(yy / 17.0)[..., None]
          + rng.randint(0, 20, size=(h, w, 3)))
    frames, boxes = [], []
    for t in range(n_frames):
        cx, cy = 90 + 5.5 * t, 80 + 1.5 * t + 6 * np.sin(t / 3.0)
        rx, ry = 22 + 0.4 * t, 34 + 0.3 * t
        mask = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0
        tex = np.stack([200 + 40 * np.sin((xx - cx) / 4.0), 120 + 60 * np.cos((yy - cy) / 5.0),
                        60 + 50 * np.sin((xx + yy) / 6.0)], axis=-1)
        f = bg.copy()
        f[mask] = tex[mask]
        frames.append(np.clip(f, 0, 255).astype(np.uint8))
        boxes.append([int(cx - rx), int(cy - ry), int(2 * rx), int(2 * ry)])
    return np.stack(frames), np.array(boxes)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_blocks, ref_box_coder, ref_fear_tracker, ref_utils = import_reference()
    from coreml_wire import decode_mlmodel
    from coreml_interp import run_graph
    from mlmodel_to_fearw import analyse, ROLE_CLS_ENCODE, ROLE_REG_ENCODE, ROLE_CLS_CORR, ROLE_REG_CORR, \
        ROLE_BBOX_TOWER, ROLE_CLS_TOWER, ROLE_BBOX_PRED, ROLE_CLS_PRED, K_NECK, K_SEP

    mlmodel = os.path.join(REF, "evaluate/MeasurePerformance/MeasurePerformance/models/FEAR-XS-NoEmbs.mlmodel")
    model = decode_mlmodel(mlmodel)
    ana = analyse(model)

    # ------------------------------------------------------------------ 1. grid + window
    gx, gy = ref_utils.make_grid(16, 16, 256)
    window = ref_fear_tracker.FEARTracker._get_tracking_window("cosine", 16)
    np.savez_compressed(os.path.join(OUT, "grid_window.npz"), grid_x=gx.numpy(), grid_y=gy.numpy(),
                        window=window.numpy())

    # ------------------------------------------------------------------ 2. geometry
    rng = np.random.RandomState(1)
    boxes = np.concatenate([rng.uniform(-40, 400, size=(64, 2)), rng.uniform(0.5, 300, size=(64, 2))], axis=1)
    boxes[:8] = np.round(boxes[:8])
    boxes = np.concatenate([boxes, np.array([[163, 53, 45, 174], [0, 0, 2, 2], [478, 254, 10, 10],
                                             [-5, -5, 3, 3], [100, 100, 1, 500]], dtype=float)])
    shape = (256, 480, 3)
    geo = dict(boxes=boxes, shape=np.array(shape))
    for off in (0.2, 2):
        geo[f"extend_{off}"] = np.stack([ref_utils.extend_bbox(b, off) for b in boxes])
    geo["ensure"] = np.stack([ref_utils.ensure_bbox_boundaries(b, shape) for b in boxes])
    geo["clamp"] = np.stack([ref_utils.clamp_bbox(b, shape) for b in boxes]).astype(np.float64)
    np.savez_compressed(os.path.join(OUT, "geometry.npz"), **geo)

    # ------------------------------------------------------------------ 3. box coder
    coder = ref_box_coder.FEARBoxCoder(tracker_config=TRACKING_CONFIG)
    g = torch.Generator().manual_seed(3)
    enc_boxes = torch.tensor([[100.0, 90.0, 60.0, 80.0], [0.0, 0.0, 256.0, 256.0], [130.5, 10.25, 20.0, 200.0],
                              [250.0, 250.0, 30.0, 30.0]], dtype=torch.float64)
    enc = coder.encode(enc_boxes)
    cls_maps = torch.randn(16, 1, 16, 16, generator=g)
    reg_maps = torch.rand(16, 4, 16, 16, generator=g) * 90 + 1
    cls_maps[3, 0, 5, 7] = cls_maps[3, 0, 9, 2] = 9.0       # exact tie -> first maximum wins
    dec_s = coder.decode(regression_map=reg_maps, classification_map=cls_maps, use_sigmoid=True)
    dec_n = coder.decode(regression_map=reg_maps, classification_map=cls_maps, use_sigmoid=False)
    np.savez_compressed(os.path.join(OUT, "box_coder.npz"), enc_boxes=enc_boxes.numpy(),
                        enc_regression=enc.regression_map.numpy(), enc_label=enc.classification_label.numpy(),
                        cls_maps=cls_maps.numpy(), reg_maps=reg_maps.numpy(),
                        dec_sigmoid_bbox=dec_s.bbox.numpy(), dec_sigmoid_rc=np.array(dec_s.pred_coords),
                        dec_plain_bbox=dec_n.bbox.numpy(), dec_plain_rc=np.array(dec_n.pred_coords))

    # ------------------------------------------------------------------ 4. tracker post-processing
    class _NoNet:
        pass

    post = {}
    g = torch.Generator().manual_seed(1)
    cls1 = torch.randn(1, 1, 16, 16, generator=g) * 2
    reg1 = torch.rand(1, 4, 16, 16, generator=g) * 60 + 5
    post["cls"], post["reg"] = cls1.numpy(), reg1.numpy()
    for smooth in (False, True):
        cfg = dict(TRACKING_CONFIG)
        if smooth:
            cfg["smooth"] = True
        trk = ref_fear_tracker.FEARTracker(_NoNet(), cuda_id="cpu", **cfg)
        trk.tracking_state.prev_size = np.array([51.2, 51.2])
        bbox, score = trk._postprocess({"TARGET_CLASSIFICATION_KEY": cls1.clone(),
                                        "TARGET_REGRESSION_LABEL_KEY": reg1.clone()})
        post[f"bbox_smooth{int(smooth)}"] = np.asarray(bbox, dtype=np.float64)
        post[f"score_smooth{int(smooth)}"] = np.asarray(float(score))
        if smooth:
            pscore, penalty = trk._confidence_postprocess(cls1.float().sigmoid(), reg1.float())
            post["pscore"], post["penalty"] = pscore.numpy(), penalty
    trk = ref_fear_tracker.FEARTracker(_NoNet(), cuda_id="cpu", **TRACKING_CONFIG)
    contexts = np.array([[100, -40, 225, 870], [73, -295, 225, 870], [0, 0, 256, 256], [-17, 33, 401, 97]])
    rs_in = np.concatenate([post["bbox_smooth1"][None], rng.uniform(0, 256, size=(15, 4)),
                            np.array([[10.5, 20.5, 2.5, 3.5], [0.5, 1.5, 0.2, 0.1]])])
    rs_out = np.stack([[trk._rescale_bbox(b.copy(), c) for c in contexts] for b in rs_in])
    post["rescale_in"], post["rescale_ctx"], post["rescale_out"] = rs_in, contexts, rs_out
    sm = [trk._smooth_size(np.array([40.0, 70.0]), np.array([51.2, 48.0]), lr) for lr in (0.1, 0.4, 0.765)]
    post["smooth_size"] = np.array(sm)
    np.savez_compressed(os.path.join(OUT, "postprocess.npz"), **post)

    # ------------------------------------------------------------------ 5. reference head modules w/ real weights
    convs = ana["convs"]

    def W(i):
        c = convs[i]
        return torch.from_numpy(c["w"].astype(np.float32)).view(c["cout"], c["cin_g"], c["k"], c["k"])

    def B(i):
        c = convs[i]
        return None if c["b"] is None else torch.from_numpy(c["b"].astype(np.float32))

    def ident_bn(bn, bias):
        # BN(x) = (x-mean)/sqrt(var+eps)*w + b  == x + bias  with mean=0, var=1-eps, w=1, b=bias
        bn.weight.data.fill_(1.0)
        bn.bias.data.copy_(bias if bias is not None else torch.zeros_like(bn.bias))
        bn.running_mean.zero_()
        bn.running_var.fill_(1.0 - bn.eps)

    def load_sep(sep, dw_i, pw_i, pw_scale=1.0, pw_bias_to_bn=None):
        sep.depthwise.weight.data.copy_(W(dw_i))
        if sep.depthwise.bias is not None:
            sep.depthwise.bias.data.copy_(B(dw_i))
        else:
            assert B(dw_i) is None
        sep.pointwise.weight.data.copy_(W(pw_i) * pw_scale)
        if sep.pointwise.bias is not None:
            if pw_bias_to_bn is None:
                sep.pointwise.bias.data.copy_(B(pw_i) * pw_scale)
            else:
                sep.pointwise.bias.data.zero_()
                ident_bn(pw_bias_to_bn, B(pw_i))
        else:
            ident_bn(pw_bias_to_bn, B(pw_i))

    blk = {}
    for b in ana["blocks"]:
        if b["kind"] == K_SEP:
            blk.setdefault(b["role"], []).append(b["conv"])
        elif b["kind"] == K_NECK:
            neck_i = b["conv"][0]
    tower = ref_blocks.BoxTower(towernum=2, inchannels=256, outchannels=256, mobile=True).eval()
    neck = ref_blocks.AdjustLayer(112, 256).eval()
    neck.downsample[0].weight.data.copy_(W(neck_i))
    ident_bn(neck.downsample[1], B(neck_i))
    load_sep(tower.cls_encode.matrix11_s[0], *blk[ROLE_CLS_ENCODE][0][:2], pw_bias_to_bn=tower.cls_encode.matrix11_s[1])
    load_sep(tower.reg_encode.matrix11_s[0], *blk[ROLE_REG_ENCODE][0][:2], pw_bias_to_bn=tower.reg_encode.matrix11_s[1])
    load_sep(tower.cls_dw.enc[0], *blk[ROLE_CLS_CORR][0][:2], pw_bias_to_bn=tower.cls_dw.enc[1])
    load_sep(tower.reg_dw.enc[0], *blk[ROLE_REG_CORR][0][:2], pw_bias_to_bn=tower.reg_dw.enc[1])
    for j, cv in enumerate(blk[ROLE_BBOX_TOWER]):
        load_sep(tower.bbox_tower[3 * j], cv[0], cv[1], pw_bias_to_bn=tower.bbox_tower[3 * j + 1])
    for j, cv in enumerate(blk[ROLE_CLS_TOWER]):
        load_sep(tower.cls_tower[3 * j], cv[0], cv[1], pw_bias_to_bn=tower.cls_tower[3 * j + 1])
    # adjust/bias folded into bbox_pred by the CoreML conversion: undo with adjust=1, bias=0;
    # the 0.1 factor on cls is applied by BoxTower.forward (blocks.py:192): pre-multiply by 10.
    load_sep(tower.bbox_pred, *blk[ROLE_BBOX_PRED][0][:2])
    tower.adjust.data.fill_(1.0)
    tower.bias.data.zero_()
    load_sep(tower.cls_pred, *blk[ROLE_CLS_PRED][0][:2], pw_scale=10.0)

    g = torch.Generator().manual_seed(5)
    trunk_out = torch.randn(3, 112, 16, 16, generator=g) * 2
    tmpl_trunk = torch.randn(3, 112, 8, 8, generator=g) * 2
    upd_trunk = torch.randn(3, 112, 8, 8, generator=g) * 2
    with torch.no_grad():
        xs = neck(trunk_out)
        zs = neck(tmpl_trunk)
        zu = neck(upd_trunk)
        bbox, cls, cls_dw, x_reg = tower(xs, zs)
        bbox_u, cls_u, _, _ = tower(xs, zs, update=zu)
    np.savez_compressed(os.path.join(OUT, "head_modules.npz"), trunk_out=trunk_out.numpy(),
                        tmpl_trunk=tmpl_trunk.numpy(), upd_trunk=upd_trunk.numpy(), neck_search=xs.numpy(),
                        neck_template=zs.numpy(), neck_update=zu.numpy(), bbox=bbox.numpy(), cls=cls.numpy(),
                        cls_dw=cls_dw.numpy(), x_reg=x_reg.numpy(), bbox_update=bbox_u.numpy(),
                        cls_update=cls_u.numpy())

    # ------------------------------------------------------------------ 6. whole-net maps (literal CoreML graph)
    def norm(u8_nchw):
        mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1) * 255.0
        inv = 1.0 / (torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1) * 255.0)
        return (u8_nchw.float() - mean) * inv

    def graph_features(x):
        z0 = torch.zeros(x.shape[0], 256, 8, 8)
        return run_graph(model, {"image": x, "template_features": z0}, keep={"input.125"}, stop_after="input.125")["input.125"]

    def graph_track(x, z):
        o = run_graph(model, {"image": x, "template_features": z}, keep={"bbox", "cls"})
        return o["bbox"], o["cls"]

    g = torch.Generator().manual_seed(0)
    search_u8 = torch.randint(0, 256, (8, 3, 256, 256), dtype=torch.uint8, generator=g)
    tmpl_u8 = torch.randint(0, 256, (8, 3, 128, 128), dtype=torch.uint8, generator=g)
    # half of the crops: structured content (smooth blobs) so the maps are not pure noise responses
    yy, xx = np.mgrid[0:256, 0:256]
    for i in range(4):
        img = np.stack([128 + 100 * np.sin(xx / (7.0 + i)) * np.cos(yy / (11.0 + 2 * i)),
                        128 + 90 * np.cos((xx + yy) / (9.0 + i)), 100 + 80 * np.sin(yy / (5.0 + i))])
        search_u8[i] = torch.from_numpy(np.clip(img, 0, 255).astype(np.uint8))
        tmpl_u8[i] = search_u8[i][:, 64:192, 64:192]
    xs_n, zs_n = norm(search_u8), norm(tmpl_u8)
    zfeat = graph_features(zs_n)
    bbox, cls = graph_track(xs_n, zfeat)
    dec = coder.decode(regression_map=bbox, classification_map=cls, use_sigmoid=True)
    flat = cls.reshape(8, -1)
    top2 = torch.topk(flat, 2, dim=1).values
    np.savez_compressed(os.path.join(OUT, "track_maps.npz"), search_u8=search_u8.numpy(), template_u8=tmpl_u8.numpy(),
                        template_features=zfeat.numpy(), search_features=graph_features(xs_n).numpy(),
                        bbox=bbox.numpy(), cls=cls.numpy(), dec_bbox=dec.bbox.numpy(),
                        dec_rc=np.array(dec.pred_coords), logit_margin=(top2[:, 0] - top2[:, 1]).numpy())

    # ------------------------------------------------------------------ 7. trunk taps (one crop)
    env = run_graph(model, {"image": xs_n[:1], "template_features": zfeat[:1]})
    taps = {}
    cur_names = []
    # block outputs in order: after each add / linear project; recover from analyse()
    for bi, b in enumerate(ana["blocks"]):
        last = [c for c in b["conv"] if c >= 0][-1]
        t = convs[last]["post_t"]
        if b["residual"]:
            # output of the add that consumes the project output
            for lay in model["layers"]:
                if lay["kind"] == "add" and lay["inputs"][0] == convs[last]["out_t"]:
                    t = lay["outputs"][0]
        if b["kind"] == K_SEP and b["act"] == 2:
            t = "bbox"
        if b["kind"] == K_SEP and b["role"] == ROLE_CLS_PRED:
            t = "cls"
        taps[f"block{bi:02d}"] = env[t].numpy()
    np.savez_compressed(os.path.join(OUT, "trunk_taps.npz"), image=xs_n[:1].numpy(),
                        template_features=zfeat[:1].numpy(), **taps)

    # ------------------------------------------------------------------ 8. synthetic clip through the reference loop
    # get_extended_crop (utils.py:215-253) needs cv2 + albumentations (absent): the reference tracker's
    # control flow runs for real, with that one function and the Normalize transform substituted by the
    # repo's numpy restatements (feartracker_amd/geometry.py).  Crop parity vs cv2 itself stays UNPINNED.
    from feartracker_amd.geometry import get_extended_crop as crop_restated, normalize_image

    class GraphNet:
        def get_features(self, x):
            return graph_features(x)

        def track(self, search, template_features):
            b, c = graph_track(search, template_features)
            return {"TARGET_REGRESSION_LABEL_KEY": b, "TARGET_CLASSIFICATION_KEY": c}

    ref_fear_tracker.get_extended_crop = crop_restated
    frames, gt = synth_clip()
    trk = ref_fear_tracker.FEARTracker(GraphNet(), cuda_id="cpu", **TRACKING_CONFIG)
    trk._template_transform = normalize_image
    trk._search_transform = normalize_image
    trk.initialize(frames[0], np.array(gt[0]))
    tracked, scores, crops, raw = [np.array(gt[0])], [], [], []
    for f in frames[1:]:
        # replicate update() but also record the crop and the raw prediction
        out = trk.update(f)
        tracked.append(np.array(out["bbox"]))
    # second pass to record intermediate tensors for the first 3 updates
    trk2 = ref_fear_tracker.FEARTracker(GraphNet(), cuda_id="cpu", **TRACKING_CONFIG)
    trk2._template_transform = normalize_image
    trk2._search_transform = normalize_image
    trk2.initialize(frames[0], np.array(gt[0]))
    tmpl_feat = trk2._template_features.numpy()
    for f in frames[1:4]:
        crop, _, ctx = crop_restated(image=f, bbox=trk2.tracking_state.bbox, crop_size=256, offset=2,
                                     padding_value=trk2.tracking_state.mean_color)
        crops.append(crop)
        pb, sc = trk2.track(crop)
        raw.append(np.asarray(pb, dtype=np.float64))
        scores.append(float(sc))
        trk2.update(f)
    np.savez_compressed(os.path.join(OUT, "clip_synth.npz"), frames=frames, init_bbox=np.array(gt[0]),
                        gt=gt, tracked=np.stack(tracked), template_features=tmpl_feat, crops=np.stack(crops),
                        raw_pred=np.stack(raw), scores=np.array(scores))
    print("golden fixtures written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print(f"  {fn}: {os.path.getsize(os.path.join(OUT, fn))} bytes")


if __name__ == "__main__":
    main()
