"""N1: the crop path against an independent statement of what OpenCV does.

cv2 exists neither here nor on the GPU box, so parity against OpenCV itself stays unpinned (`if cv2:` branches below run the
day it is importable).  What IS pinned: the product's two crop implementations — feartracker_amd/geometry.py (vectorised numpy)
and fear_crop_normalize (HIP, one thread per output pixel) — reproduce, byte for byte,
  * oracle/cv_ref.c: resize.cpp's 8u INTER_LINEAR restated table-driven (xofs / ialpha / yofs / ibeta, HResizeLinear,
    VResizeLinear) by a separate path, on a sweep of sizes and on known answers that follow from the algorithm's definition;
  * tests/golden/crop_cv.npz: the REFERENCE's own get_extended_crop (utils.py:215-253), run unmodified in the build container
    on those restatements (tools/make_golden.py section 9b): the demo context [73,-295,225,870], out-of-frame boxes on every
    side, the identity-size and exact-2x special cases of cv::resize.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from feartracker_amd import geometry as geo
from oracle import cv_ref

try:
    import cv2
except ImportError:
    cv2 = None


def _cases(golden_dir):
    d = np.load(f"{golden_dir}/crop_cv.npz")
    from clipgen import demo_clip, frame_crcs
    frames, _ = demo_clip()
    idx = d["frame_index"]
    np.testing.assert_array_equal(frame_crcs(frames[idx]), d["frame_crc32"])          # the generator has not drifted
    for i, name in enumerate(d["case_names"]):
        yield str(name), frames[idx[i]], d["bbox"][i], int(d["crop_size"][i]), float(d["offset"][i]), d


def test_oracle_resize_known_answers():
    """Properties that follow from the definition of the fixed-point bilinear resize, not from anyone's memory of it."""
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, size=(41, 67, 3)).astype(np.uint8)
    np.testing.assert_array_equal(cv_ref.resize_linear_u8(img, 41, 67), img)                          # identity
    const = np.full((23, 31, 3), 201, np.uint8)
    assert np.all(cv_ref.resize_linear_u8(const, 256, 256) == 201)                                     # weights sum to 2048
    big = rng.randint(0, 256, size=(64, 96, 3)).astype(np.int64)
    box = ((big[0::2, 0::2] + big[0::2, 1::2] + big[1::2, 0::2] + big[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    np.testing.assert_array_equal(cv_ref.resize_linear_u8(big.astype(np.uint8), 32, 48), box)          # exact 2x = box mean
    # integer upscale x2 of a horizontal ramp: sample positions (dx + .5) / 2 - .5 = -.25, .25, .75, 1.25 ... -> weights
    # (2048, 0) clamped, then alternately (1536, 512) and (512, 1536)
    ramp = np.tile((np.arange(8) * 16).astype(np.uint8)[None, :, None], (4, 1, 1))
    up = cv_ref.resize_linear_u8(ramp, 8, 16)[0, :, 0].astype(int)
    want = [0] + [((16 * k) * w0 + (16 * (k + 1)) * w1 + 1024) // 2048 for k in range(7) for (w0, w1) in ((1536, 512), (512, 1536))] + [112]
    assert up.tolist() == want
    idx, w0, w1 = cv_ref.linear_table(16, 8, True)
    assert idx.tolist() == [0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7]
    assert w1.tolist() == [0, 512, 1536, 512, 1536, 512, 1536, 512, 1536, 512, 1536, 512, 1536, 512, 1536, 0]
    # rows are not clamped in the table: the first destination row of a 2x upscale sits at -0.25
    idy, b0, b1 = cv_ref.linear_table(16, 8, False)
    assert idy[0] == -1 and (b0[0], b1[0]) == (512, 1536) and idy[-1] == 7 and (b0[-1], b1[-1]) == (1536, 512)
    b = cv_ref.copy_make_border_constant(img, 2, 3, 4, 5, np.array([10.5, 11.5, 300.0]))
    assert b.shape == (46, 76, 3) and b[0, 0].tolist() == [10, 12, 255] and b[-1, -1].tolist() == [10, 12, 255]
    np.testing.assert_array_equal(b[2:43, 4:71], img)


def test_host_resize_is_bit_identical_to_the_table_driven_restatement():
    rng = np.random.RandomState(0)
    sizes = [(rng.randint(1, 900), rng.randint(1, 900)) for _ in range(260)]
    sizes += [(512, 512), (256, 256), (255, 257), (1, 1), (2, 3), (870, 225), (225, 870), (128, 128), (64, 2000), (1080, 12)]
    for i, (sh, sw) in enumerate(sizes):
        out = (256, 128, 255)[i % 3]
        img = rng.randint(0, 256, size=(sh, sw, 3)).astype(np.uint8)
        np.testing.assert_array_equal(geo.resize_bilinear_u8(img, out, out), cv_ref.resize_linear_u8(img, out, out),
                                      err_msg=f"{sh}x{sw} -> {out}")
    for dst, src in ((256, 870), (256, 225), (128, 63), (128, 243), (256, 256), (256, 1), (255, 1000)):
        for clamp in (True, False):
            a = geo._linear_taps(dst, src, clamp)
            b = cv_ref.linear_table(dst, src, clamp)
            for x, y in zip(a, b):
                np.testing.assert_array_equal(np.asarray(x, np.int64), np.asarray(y, np.int64))
    img = rng.randint(0, 256, size=(37, 53, 3)).astype(np.uint8)
    val = np.array([99.5, 100.5, -3.0])
    np.testing.assert_array_equal(geo.copy_make_border(img, 5, 0, 7, 2, val), cv_ref.copy_make_border_constant(img, 5, 0, 7, 2, val))


def test_host_crop_reproduces_the_reference_crops(golden_dir):
    """geometry.get_extended_crop vs the reference's get_extended_crop on the OpenCV restatement: crop bytes, context box and
    the float64 box inside the crop (albumentations' own sequence), every case."""
    n = 0
    for name, frame, box, size, off, d in _cases(golden_dir):
        mean = np.mean(frame, axis=(0, 1))
        np.testing.assert_array_equal(mean, d[f"pad_{name}"])
        crop, box_in_crop, ctx = geo.get_extended_crop(frame, box, size, off, padding_value=mean)
        np.testing.assert_array_equal(ctx, d[f"ctx_{name}"], err_msg=name)
        np.testing.assert_array_equal(crop, d[f"crop_{name}"], err_msg=name)
        np.testing.assert_array_equal(box_in_crop, d[f"box_{name}"], err_msg=name)
        ctx2, box2 = geo.crop_geometry(frame.shape, box, size, off)
        np.testing.assert_array_equal(ctx2, ctx)
        np.testing.assert_array_equal(box2, box_in_crop)
        if cv2 is not None:                                                          # the real thing, whenever it exists
            pl, pt = max(-ctx[0], 0), max(-ctx[1], 0)
            pr, pb = max(ctx[0] + ctx[2] - frame.shape[1], 0), max(ctx[1] + ctx[3] - frame.shape[0], 0)
            inner = frame[ctx[1] + pt: ctx[1] + ctx[3] - pb, ctx[0] + pl: ctx[0] + ctx[2] - pr]
            padded = cv2.copyMakeBorder(inner, pt, pb, pl, pr, cv2.BORDER_CONSTANT, value=mean)
            np.testing.assert_array_equal(cv2.resize(padded, (size, size), interpolation=cv2.INTER_LINEAR), crop, err_msg=name)
        n += 1
    assert n == 10
    assert tuple(np.load(f"{golden_dir}/crop_cv.npz")["ctx_demo_search"]) == (73, -295, 225, 870)


@pytest.mark.gpu
def test_device_crop_reproduces_the_reference_crops(golden_dir):
    """fear_crop_normalize (through the C ABI) vs the same fixture: the normalised floats of every reference crop, bit for
    bit (normalisation itself is pinned by the CoreML scaler fixture)."""
    import torch
    from conftest import WEIGHTS
    from feartracker_amd import FEARNetHIP
    net = FEARNetHIP(WEIGHTS, device=0, max_batch=4)
    for name, frame, box, size, off, d in _cases(golden_dir):
        mean = d[f"pad_{name}"]
        want = np.transpose(geo.normalize_image(d[f"crop_{name}"]), (2, 0, 1))
        got = net.crop_normalize(torch.from_numpy(frame).cuda(), d[f"ctx_{name}"], geo.border_color_u8(mean), size)[0].cpu().numpy()
        np.testing.assert_array_equal(got, want, err_msg=name)
        # the same crop from a HOST frame (only the context rectangle is uploaded, boxes shifted to it), given as a numpy view
        # with a negative stride (a BGR frame flipped to RGB) and as a CPU tensor
        bgr = np.ascontiguousarray(frame[:, :, ::-1])
        for host in (bgr[:, :, ::-1], torch.from_numpy(frame)):
            got = net.crop_normalize(host, d[f"ctx_{name}"], geo.border_color_u8(mean), size)[0].cpu().numpy()
            np.testing.assert_array_equal(got, want, err_msg=name + " (host frame)")


@pytest.mark.gpu
def test_device_crop_matches_the_table_driven_restatement_on_a_size_sweep():
    """Contexts of many sizes and positions over one frame (in, across and outside it): device crop == C restatement."""
    import torch
    from conftest import WEIGHTS
    from clipgen import demo_clip
    from feartracker_amd import FEARNetHIP
    net = FEARNetHIP(WEIGHTS, device=0, max_batch=4)
    frame = demo_clip(n_frames=3)[0][2]
    fr = torch.from_numpy(frame).cuda()
    mean = np.mean(frame, axis=(0, 1))
    rng = np.random.RandomState(21)
    for i in range(60):
        w, h = rng.randint(3, 420), rng.randint(3, 300)
        # what the tracker can hold: initialize()/update() clamp the box to the frame (fear_tracker.py:24,62), so a context
        # always overlaps it (a context wholly outside makes the reference's own slice `image[a:-b]` wrap around)
        box = geo.clamp_bbox(np.array([rng.randint(-60, 470), rng.randint(-60, 250), w, h]), frame.shape).astype(np.float64)
        size, off = ((256, 2.0), (128, 0.2), (256, 0.1))[i % 3]
        crop, ctx = cv_ref.get_extended_crop(frame, box, size, off, padding_value=mean)
        want = np.transpose(geo.normalize_image(crop), (2, 0, 1))
        got = net.crop_normalize(fr, ctx, geo.border_color_u8(mean), size)[0].cpu().numpy()
        np.testing.assert_array_equal(got, want, err_msg=f"case {i}: box {box.tolist()} -> {size}")
