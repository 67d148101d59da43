"""Host-side box geometry and crop extraction for the FEAR per-frame path.

Mirrors the *behaviour* (same names, argument meaning, integer truncation and rounding
rules) of the reference helpers in model_training/utils/utils.py so that the tracker state
evolves identically:

* `extend_bbox`            utils.py:29-57   (int32 truncation toward zero at :57)
* `ensure_bbox_boundaries` utils.py:60-71
* `clamp_bbox`             utils.py:202-212
* `get_extended_crop`      utils.py:215-253 (pad with mean colour, anisotropic resize)
* `make_grid`              utils.py:184-199 (float64 grids)

The reference delegates padding to `cv2.copyMakeBorder` and resizing to albumentations'
`A.Resize` (= `cv2.resize(..., INTER_LINEAR)`); neither library exists in this image, so
`copy_make_border` / `resize_bilinear_u8` below restate OpenCV's uint8 behaviour
(saturate-cast of the border value; 11-bit fixed-point bilinear with half-pixel centres).
Crop parity against real cv2 is UNPINNED here; what pins these functions is the
independent, table-driven C restatement of resize.cpp that the tests carry (cv_ref.c), bit for bit.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple, Union

import numpy as np

_COEF_BITS = 11                      # OpenCV INTER_RESIZE_COEF_BITS
_COEF_ONE = 1 << _COEF_BITS


def extend_bbox(bbox: Sequence[float], offset: Union[float, Tuple[float, ...]] = 0.1) -> np.ndarray:
    """Grow an xywh box by `offset` x its size on every side; result truncated to int32."""
    x, y, w, h = bbox
    if isinstance(offset, tuple):
        if len(offset) == 4:
            o_l, o_r, o_t, o_b = offset
        elif len(offset) == 2:
            o_l = o_r = offset[0]
            o_t = o_b = offset[1]
        else:
            raise ValueError("offset tuple must have 2 or 4 entries")
    else:
        o_l = o_r = o_t = o_b = offset
    grown = np.array([x - w * o_l, y - h * o_t, w * (1.0 + o_r + o_l), h * (1.0 + o_t + o_b)])
    return grown.astype("int32")


def ensure_bbox_boundaries(bbox: Sequence[float], img_shape: Tuple[int, ...]) -> np.ndarray:
    """Intersect an xywh box with the image rectangle (img_shape = (H, W, ...)); int32 result."""
    img_h, img_w = img_shape[0], img_shape[1]
    x, y, w, h = bbox
    x_lo = min(max(0, x), img_w)
    y_lo = min(max(0, y), img_h)
    x_hi = min(max(0, x_lo + w), img_w)
    y_hi = min(max(0, y_lo + h), img_h)
    return np.array([x_lo, y_lo, x_hi - x_lo, y_hi - y_lo]).astype("int32")


def clamp_bbox(bbox: Sequence[float], shape: Tuple[int, ...], min_side: int = 3) -> np.ndarray:
    """Clip to the frame, then enforce a minimum side, shifting the origin back inside."""
    x, y, w, h = ensure_bbox_boundaries(bbox, img_shape=shape)
    img_h, img_w = shape[0], shape[1]
    if w < min_side:
        w = min_side
        x -= max(0, x + w - img_w)
    if h < min_side:
        h = min_side
        y -= max(0, y + h - img_h)
    return np.array([x, y, w, h])


def make_grid(score_size: int, total_stride: int, instance_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Centres of the score-map cells in search-crop pixels, float64, shape (1, S, S) each."""
    ticks = (np.arange(0, score_size) - np.floor(float(score_size // 2))) * total_stride + instance_size // 2
    gx, gy = np.meshgrid(ticks, ticks)
    return gx[np.newaxis, :, :], gy[np.newaxis, :, :]


def _saturate_u8(values: np.ndarray) -> np.ndarray:
    """cv::saturate_cast<uchar>(double): round half to even, clamp to [0, 255]."""
    return np.clip(np.rint(values), 0, 255).astype(np.uint8)


def copy_make_border(img: np.ndarray, top: int, bottom: int, left: int, right: int,
                     value: Sequence[float]) -> np.ndarray:
    """Constant border (cv2.BORDER_CONSTANT) with a per-channel value saturate-cast to uint8."""
    h, w = img.shape[:2]
    c = img.shape[2] if img.ndim == 3 else 1
    fill = _saturate_u8(np.asarray(value, dtype=np.float64).reshape(-1)[:c])
    out = np.empty((h + top + bottom, w + left + right, c), dtype=np.uint8)
    out[...] = fill
    out[top:top + h, left:left + w] = img.reshape(h, w, c)
    return out


def _linear_taps(dst: int, src: int, clamp: bool) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Source index and the two int16 fixed-point weights per destination coordinate, in OpenCV's order of operations:
    the position is rounded to float32 FIRST, floored, and the fraction is the float32 difference (resize.cpp: `fx =
    (float)((dx+0.5)*scale_x - 0.5); sx = cvFloor(fx); fx -= sx`), scale = 1 / (dst / src) in double.  Columns
    (`clamp=True`) pin positions left of pixel 0 / right of the last pixel to that pixel with weight 2048 | 0; rows
    (`clamp=False`) keep their raw index and weights — the caller clips the two ROW INDICES instead, so an edge row is
    blended with itself at the table's weights, which is not the same number after the >> 16 truncations."""
    scale = 1.0 / (float(dst) / float(src))
    pos = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    idx_f = np.floor(pos)
    frac = pos - idx_f                                   # float32 - float32
    idx = idx_f.astype(np.int64)
    if clamp:
        low = idx < 0
        frac[low] = 0.0
        idx[low] = 0
        high = idx >= src - 1
        frac[high] = 0.0
        idx[high] = src - 1
    w0 = np.rint((np.float32(1.0) - frac) * np.float32(_COEF_ONE)).astype(np.int64)     # cvRound: half to even
    w1 = np.rint(frac * np.float32(_COEF_ONE)).astype(np.int64)
    return idx, w0, w1


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC.

    Horizontal pass in 11-bit fixed point to int32, vertical pass
    `(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2` as in OpenCV's 8u linear resizer; an exact 2x2 decimation is
    the 2x2 box mean `(a+b+c+d+2)>>2` (cv::resize runs INTER_LINEAR as the fast INTER_AREA there).  Compared bit for bit
    with the independent table-driven restatement cv_ref.c in tests/test_cv_parity.py.
    """
    if img.dtype != np.uint8:
        raise TypeError("resize_bilinear_u8 expects uint8 input")
    src_h, src_w = img.shape[:2]
    if (src_h, src_w) == (out_h, out_w):
        return img.copy()
    src = img.astype(np.int64)
    if src_h == 2 * out_h and src_w == 2 * out_w:
        return ((src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    ix, ax0, ax1 = _linear_taps(out_w, src_w, clamp=True)
    iy, ay0, ay1 = _linear_taps(out_h, src_h, clamp=False)
    ix1 = np.minimum(ix + 1, src_w - 1)
    iy0 = np.clip(iy, 0, src_h - 1)
    iy1 = np.clip(iy + 1, 0, src_h - 1)
    rows = src[:, ix] * ax0[None, :, None] + src[:, ix1] * ax1[None, :, None]      # (src_h, out_w, C)
    s0 = rows[iy0] >> 4
    s1 = rows[iy1] >> 4
    out = (((ay0[:, None, None] * s0) >> 16) + ((ay1[:, None, None] * s1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def crop_geometry(image_shape: Tuple[int, ...], bbox: Sequence[float], crop_size: int,
                  offset: float) -> Tuple[np.ndarray, np.ndarray]:
    """The pixel-free part of `get_extended_crop`: (context box int32 xywh in frame coordinates, bbox inside the
    resized crop).  Used by the device crop path (`FEARNetHIP.crop_normalize`), which does the pixel work on the GPU."""
    img_h, img_w = image_shape[0], image_shape[1]
    ctx = extend_bbox(bbox, offset)
    cx, cy, cw, ch = (int(v) for v in ctx)
    box_in_pad = ensure_bbox_boundaries(
        np.array([bbox[0] - ctx[0], bbox[1] - ctx[1], bbox[2], bbox[3]]), img_shape=(ch, cw))
    return ctx, _resized_coco_box(box_in_pad, ch, cw, crop_size)


def _resized_coco_box(box: Sequence[float], rows: int, cols: int, size: int) -> np.ndarray:
    """An xywh box of a rows x cols image after A.Compose([A.Resize(size, size)], bbox_params=coco) — albumentations'
    own float64 sequence (bbox_utils: corners / (cols, rows), clip to [0, 1], * size, width = x_max - x_min)."""
    x_min, y_min = box[0] / cols, box[1] / rows
    x_max, y_max = (box[0] + box[2]) / cols, (box[1] + box[3]) / rows
    x_min, y_min, x_max, y_max = (min(max(float(v), 0.0), 1.0) for v in (x_min, y_min, x_max, y_max))
    x_min, x_max, y_min, y_max = x_min * size, x_max * size, y_min * size, y_max * size
    return np.array([x_min, y_min, x_max - x_min, y_max - y_min])


def border_color_u8(padding_value: Sequence[float]) -> np.ndarray:
    """The uint8 colour cv2.copyMakeBorder writes for a float border value (round half to even, saturate)."""
    return _saturate_u8(np.asarray(padding_value, dtype=np.float64).reshape(-1)[:3])


def get_extended_crop(image: np.ndarray, bbox: Sequence[float], crop_size: int, offset: float,
                      padding_value: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Context crop around `bbox`, padded with `padding_value`, resized to crop_size x crop_size.

    Returns (crop uint8 HxWx3, bbox inside the crop (xywh, crop pixels), context box int32 xywh
    in frame coordinates) like the reference's utils.py:215-253.
    """
    if padding_value is None:
        padding_value = np.mean(image, axis=(0, 1))
    img_h, img_w = image.shape[:2]
    ctx = extend_bbox(bbox, offset)
    cx, cy, cw, ch = (int(v) for v in ctx)
    pad_l, pad_t = max(-cx, 0), max(-cy, 0)
    pad_r, pad_b = max(cx + cw - img_w, 0), max(cy + ch - img_h, 0)
    inner = image[cy + pad_t: cy + ch - pad_b, cx + pad_l: cx + cw - pad_r]
    padded = copy_make_border(inner, pad_t, pad_b, pad_l, pad_r, padding_value)
    box_in_pad = ensure_bbox_boundaries(
        np.array([bbox[0] - ctx[0], bbox[1] - ctx[1], bbox[2], bbox[3]]), img_shape=padded.shape[:2])
    crop = resize_bilinear_u8(padded, crop_size, crop_size)
    ph, pw = padded.shape[:2]
    return crop, _resized_coco_box(box_in_pad, ph, pw, crop_size), ctx


_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32) * np.float32(255.0)
_INV_STD = np.reciprocal(np.array([0.229, 0.224, 0.225], dtype=np.float32) * np.float32(255.0), dtype=np.float32)


def normalize_image(img_hwc_u8: np.ndarray) -> np.ndarray:
    """ImageNet normalisation of an RGB uint8 HxWx3 image -> fp32 HxWx3:
    (px - 255*mean) * (1 / (255*std)), the arithmetic of base_tracker.py:70-81."""
    out = img_hwc_u8.astype(np.float32)
    out -= _MEAN
    out *= _INV_STD
    return out
