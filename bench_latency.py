#!/usr/bin/env python3
"""Batch-1 latency of the drop-in tracker loop (BASELINE.json configs[0] / SURVEY.md §8d "Config 1").

`initialize` on frame 0 + `update` on every following frame, exactly the loop of demo_video.track
(demo_video.py:22-28), on the deterministic synthetic clip of tests/golden/clip_synth.npz (the image has no H.264
decoder, so assets/test.mp4 cannot be read; pass --video with a .npy/.npz `frames` array of a decoded clip to use
one).  Reports ms/frame split into host crop+resize, normalise+H2D, network (`net.track`), decode+rescale, for the
HIP engine and for the CPU oracle (torch fp32, cuda_id="cpu") on the same frames, and checks that both produce the
same boxes.  One JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from feartracker_amd import DEFAULT_TRACKING_CONFIG, DEFAULT_WEIGHTS, FEARTracker  # noqa: E402
from feartracker_amd import geometry as geo  # noqa: E402


def run(tracker, frames, init_box, repeats, sync):
    t = dict(crop=0.0, pre=0.0, net=0.0, post=0.0)
    boxes = []
    n = 0
    for rep in range(repeats):
        tracker.initialize(frames[0], np.array(init_box))
        cfg, st = tracker.tracking_config, tracker.tracking_state
        for f in frames[1:]:
            t0 = time.perf_counter()
            crop, box_in_crop, ctx = geo.get_extended_crop(f, st.bbox, cfg["instance_size"], cfg["search_context"],
                                                           padding_value=st.mean_color)
            st.mapping, st.prev_size = ctx, box_in_crop[2:]
            t1 = time.perf_counter()
            x = tracker._preprocess_image(crop, tracker._search_transform)
            sync()
            t2 = time.perf_counter()
            out = tracker.net.track(x, tracker._template_features)
            sync()
            t3 = time.perf_counter()
            pred, _ = tracker._postprocess(out)
            pred = geo.clamp_bbox(tracker._rescale_bbox(pred, st.mapping), f.shape)
            st.bbox = pred
            t4 = time.perf_counter()
            if rep > 0 or repeats == 1:
                t["crop"] += t1 - t0; t["pre"] += t2 - t1; t["net"] += t3 - t2; t["post"] += t4 - t3
                n += 1
            if rep == 0:
                boxes.append(np.array(pred))
    return {k: 1e3 * v / max(n, 1) for k, v in t.items()}, np.stack(boxes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=5, help="passes over the clip (the first is warm-up when > 1)")
    ap.add_argument("--video", default=os.path.join(ROOT, "tests", "golden", "clip_synth.npz"))
    ap.add_argument("--math", type=int, default=0)
    args = ap.parse_args()
    d = np.load(args.video)
    frames = d["frames"]
    init_box = d["init_bbox"] if "init_bbox" in d else np.array([163, 53, 45, 174])

    from feartracker_amd import FEARNetHIP
    net = FEARNetHIP(DEFAULT_WEIGHTS, device=0, max_batch=1)
    net.set_math(args.math)
    hip = FEARTracker(net, cuda_id=0, **dict(DEFAULT_TRACKING_CONFIG, device_crop=False, device_postprocess=False))
    hip_ms, hip_boxes = run(hip, frames, init_box, args.repeats, torch.cuda.synchronize)

    # device crop path (fear_crop_normalize) and device crop + device post-processing (fear_decode): whole update() timed,
    # frame uploaded per call
    def run_device(**opts):
        trk = FEARTracker(net, cuda_id=0, **dict(DEFAULT_TRACKING_CONFIG, **opts))
        boxes, t_sum, n_sum = [], 0.0, 0
        for rep in range(args.repeats):
            trk.initialize(frames[0], np.array(init_box))
            for f in frames[1:]:
                t0 = time.perf_counter()
                b = trk.update(f)["bbox"]
                torch.cuda.synchronize()
                if rep > 0 or args.repeats == 1:
                    t_sum += time.perf_counter() - t0
                    n_sum += 1
                if rep == 0:
                    boxes.append(np.array(b))
        return boxes, t_sum, n_sum

    dev_boxes, dev_t, dev_n = run_device(device_crop=True)
    dpp_boxes, dpp_t, dpp_n = run_device(device_crop=True, device_postprocess=True)

    from oracle.fear_oracle import OracleNet  # CPU baseline leg only
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cpu = FEARTracker(OracleNet(DEFAULT_WEIGHTS), cuda_id="cpu", **DEFAULT_TRACKING_CONFIG)
    cpu_ms, cpu_boxes = run(cpu, frames, init_box, 2, lambda: None)

    total = sum(hip_ms.values())
    print(json.dumps({
        "metric": "ms per frame, batch-1 FEARTracker.update (FEAR-XS)", "value": total, "unit": "ms/frame",
        "fps": 1e3 / total, "higher_is_better": False, "frames": int(len(frames) - 1), "frame_shape": list(frames[0].shape),
        "split_ms": hip_ms, "cpu_oracle_split_ms": cpu_ms, "cpu_oracle_ms_per_frame": sum(cpu_ms.values()),
        "device_crop_ms_per_frame": 1e3 * dev_t / max(dev_n, 1),
        "device_crop_boxes_identical": bool(np.array_equal(np.stack(dev_boxes), hip_boxes)),
        "device_crop_and_postprocess_ms_per_frame": 1e3 * dpp_t / max(dpp_n, 1),
        "device_crop_and_postprocess_boxes_identical": bool(np.array_equal(np.stack(dpp_boxes), hip_boxes)),
        "cpu_threads": torch.get_num_threads(), "boxes_identical_to_cpu_oracle": bool(np.array_equal(hip_boxes, cpu_boxes)),
        "data": os.path.basename(args.video), "math": args.math}))


if __name__ == "__main__":
    main()
