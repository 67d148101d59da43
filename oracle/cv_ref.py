"""ctypes front end of oracle/cv_ref.c (OpenCV's 8u INTER_LINEAR resize and constant copyMakeBorder, restated) and the
reference's `get_extended_crop` (model_training/utils/utils.py:215-253) written on top of it.  TEST INFRASTRUCTURE ONLY:
only tests/, tools/make_golden.py and __graft_entry__ (which builds it) touch this module; nothing under feartracker_amd/
imports it.  "Parity unpinned" against real OpenCV — cv2 exists neither in the build image nor on the GPU box; see the
header of cv_ref.c for what is restated and from where.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_DIR, "cv_ref.c")
_LIB = os.path.join(_DIR, "libcvref.so")
_handle = None


def build(force: bool = False) -> str:
    """gcc -O2 -shared oracle/cv_ref.c -> oracle/libcvref.so (git-ignored; travels to the GPU box with the snapshot)."""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-Wall", "-Wextra", "-o", _LIB + ".tmp", _SRC, "-lm"], check=True)
        os.replace(_LIB + ".tmp", _LIB)
    return _LIB


def _lib():
    global _handle
    if _handle is None:
        h = ctypes.CDLL(build())
        u8p, i32p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int)
        h.cvref_resize_linear_8u.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long, u8p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_long]
        h.cvref_copy_make_border_const_8u.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long, u8p,
                                                      ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                      ctypes.POINTER(ctypes.c_double)]
        h.cvref_linear_table.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, i32p, i32p, i32p]
        _handle = h
    return _handle


def _u8(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def resize_linear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR) for a uint8 HxWxC (or HxW) image."""
    if img.dtype != np.uint8:
        raise TypeError("uint8 only")
    src = np.ascontiguousarray(img if img.ndim == 3 else img[:, :, None])
    sh, sw, cn = src.shape
    dst = np.empty((out_h, out_w, cn), np.uint8)
    rc = _lib().cvref_resize_linear_8u(_u8(src), sh, sw, cn, sw * cn, _u8(dst), out_h, out_w, out_w * cn)
    if rc != 0:
        raise ValueError("cvref_resize_linear_8u rejected its arguments")
    return dst if img.ndim == 3 else dst[:, :, 0]


def copy_make_border_constant(img: np.ndarray, top: int, bottom: int, left: int, right: int, value: Sequence[float]) -> np.ndarray:
    """cv2.copyMakeBorder(img, top, bottom, left, right, cv2.BORDER_CONSTANT, value=value) for uint8 HxWxC."""
    src = np.ascontiguousarray(img)
    sh, sw, cn = src.shape
    val = (ctypes.c_double * 4)(*([float(v) for v in np.asarray(value, np.float64).reshape(-1)[:4]] + [0.0] * 4)[:4])
    dst = np.empty((sh + top + bottom, sw + left + right, cn), np.uint8)
    rc = _lib().cvref_copy_make_border_const_8u(_u8(src), sh, sw, cn, sw * cn, _u8(dst), dst.shape[1] * cn, top, bottom, left,
                                                right, val)
    if rc != 0:
        raise ValueError("cvref_copy_make_border_const_8u rejected its arguments")
    return dst


def linear_table(dst: int, src: int, clamp: bool) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(source index, weight of it, weight of the next one) per destination coordinate; clamp=True = the column rule."""
    idx, w0, w1 = (np.empty(dst, np.int32) for _ in range(3))
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    if _lib().cvref_linear_table(dst, src, int(clamp), p(idx), p(w0), p(w1)) != 0:
        raise ValueError("cvref_linear_table rejected its arguments")
    return idx, w0, w1


def get_extended_crop(image: np.ndarray, bbox: Sequence[float], crop_size: int, offset: float,
                      padding_value: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    """The pixel side of the reference's get_extended_crop (utils.py:215-253): (crop uint8 SxSx3, context int32 xywh).
    extend_bbox (utils.py:29-57, scalar offset) is restated inline: grow by offset*size per side, truncate toward zero."""
    if padding_value is None:
        padding_value = np.mean(image, axis=(0, 1))
    x, y, w, h = (float(v) for v in bbox)
    ctx = np.array([x - w * offset, y - h * offset, w * (1 + 2 * offset), h * (1 + 2 * offset)]).astype("int32")
    pad_l, pad_t = max(-int(ctx[0]), 0), max(-int(ctx[1]), 0)
    pad_r = max(int(ctx[0]) + int(ctx[2]) - image.shape[1], 0)
    pad_b = max(int(ctx[1]) + int(ctx[3]) - image.shape[0], 0)
    inner = image[int(ctx[1]) + pad_t: int(ctx[1]) + int(ctx[3]) - pad_b, int(ctx[0]) + pad_l: int(ctx[0]) + int(ctx[2]) - pad_r]
    padded = copy_make_border_constant(inner, pad_t, pad_b, pad_l, pad_r, padding_value)
    return resize_linear_u8(padded, crop_size, crop_size), ctx


# ---------------------------------------------------------------------------------------------------------------------
# albumentations 1.0.0 (requirements.txt:11), the three pieces the reference's per-frame path calls, restated so that the
# reference's OWN get_extended_crop / _get_default_transform can run in the build container (tools/make_golden.py installs
# these as the `albumentations` / `cv2` modules the reference imports):
#   A.Resize(h, w)                   functional.resize: identity when the size matches, else cv2.resize(..., INTER_LINEAR)
#   A.Normalize(mean, std)           functional.normalize: fp32  (img - mean*255) * reciprocal(std*255)
#   A.Compose(..., bbox_params=coco) bbox_utils: coco -> (x_min, y_min, x_max, y_max) / (cols, rows) -> transform (Resize
#                                    leaves normalised boxes alone) -> filter_bboxes (clip to [0, 1], drop empty) ->
#                                    * (new cols, rows) -> (x_min, y_min, x_max - x_min, y_max - y_min), all float64
class AlbuResize:
    def __init__(self, height: int, width: int):
        self.height, self.width = int(height), int(width)

    def apply(self, img: np.ndarray) -> np.ndarray:
        if img.shape[:2] == (self.height, self.width):
            return img
        return resize_linear_u8(img, self.height, self.width)


class AlbuNormalize:
    def __init__(self, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), max_pixel_value: float = 255.0):
        self.mean = np.array(mean, dtype=np.float32)
        self.mean *= max_pixel_value
        std = np.array(std, dtype=np.float32)
        std *= max_pixel_value
        self.denominator = np.reciprocal(std, dtype=np.float32)

    def apply(self, img: np.ndarray) -> np.ndarray:
        out = img.astype(np.float32)
        out -= self.mean
        out *= self.denominator
        return out


class AlbuCompose:
    def __init__(self, transforms, bbox_params=None):
        self.transforms = list(transforms)
        self.bbox_params = bbox_params
        if bbox_params is not None and bbox_params.get("format") != "coco":
            raise NotImplementedError("only the coco format the reference uses")

    def __call__(self, **data):
        img = data["image"]
        out = dict(data)
        boxes = None
        if self.bbox_params is not None and "bboxes" in data:
            rows, cols = img.shape[:2]
            boxes = []
            for b in data["bboxes"]:
                x_min, y_min, w, h = (b[0], b[1], b[2], b[3])
                x_max, y_max = x_min + w, y_min + h
                nb = (x_min / cols, y_min / rows, x_max / cols, y_max / rows)
                for name, v in zip(("x_min", "y_min", "x_max", "y_max"), nb):          # check_bbox
                    if not 0 <= v <= 1:
                        raise ValueError(f"Expected {name} for bbox {nb} to be in the range [0.0, 1.0], got {v}.")
                if nb[2] <= nb[0] or nb[3] <= nb[1]:
                    raise ValueError(f"bbox {nb}: max must exceed min")
                boxes.append(nb)
        for t in self.transforms:
            img = t.apply(img)
        out["image"] = img
        if boxes is not None:
            rows, cols = img.shape[:2]
            kept = []
            for nb in boxes:
                area = (nb[2] - nb[0]) * cols * (nb[3] - nb[1]) * rows
                cb = tuple(np.clip(nb, 0, 1.0))
                carea = (cb[2] - cb[0]) * cols * (cb[3] - cb[1]) * rows
                if not area or carea / area <= self.bbox_params.get("min_visibility", 0.0):
                    continue
                if carea <= self.bbox_params.get("min_area", 0.0):
                    continue
                x_min, x_max = cb[0] * cols, cb[2] * cols
                y_min, y_max = cb[1] * rows, cb[3] * rows
                kept.append((x_min, y_min, x_max - x_min, y_max - y_min))
            out["bboxes"] = kept
        return out


def install_reference_stubs():
    """Put `cv2` and `albumentations` modules into sys.modules that carry exactly what the reference's per-frame path calls
    (cv2.copyMakeBorder + BORDER_CONSTANT; A.Compose / A.Resize / A.Normalize), backed by the restatements above.  Used by
    tools/make_golden.py in the build container so that the reference's own utils.get_extended_crop runs unmodified."""
    import sys
    import types

    class _Unused:                      # anything else the reference touches at import time (augmentation tables, ...)
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Unused()

        def __getattr__(self, name):
            return _Unused()

    class _Stub(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _Unused

    cv2 = _Stub("cv2")
    cv2.__path__ = []
    cv2.BORDER_CONSTANT = 0
    cv2.INTER_LINEAR = 1

    def _copy_make_border(src, top, bottom, left, right, borderType, dst=None, value=None):
        if borderType != cv2.BORDER_CONSTANT:
            raise NotImplementedError
        return copy_make_border_constant(src, int(top), int(bottom), int(left), int(right), value)

    def _resize(src, dsize, dst=None, fx=0, fy=0, interpolation=1):
        if interpolation != cv2.INTER_LINEAR:
            raise NotImplementedError
        return resize_linear_u8(src, int(dsize[1]), int(dsize[0]))

    cv2.copyMakeBorder = _copy_make_border
    cv2.resize = _resize
    albu = _Stub("albumentations")
    albu.__path__ = []
    albu.Compose, albu.Resize, albu.Normalize = AlbuCompose, AlbuResize, AlbuNormalize
    sys.modules["cv2"] = cv2
    sys.modules["albumentations"] = albu
    return cv2, albu
