"""Drop-in `FEARTracker` for the MI355X engine.

API mirror of the reference tracker (model_training/tracker/base_tracker.py:28-205,
model_training/tracker/fear_tracker.py:13-86): same constructor signature
`FEARTracker(model, cuda_id=0, **tracking_config)` (what `hydra.utils.instantiate(
config["tracker"], model=model)` calls, demo_video.py:18), same `initialize / update / track /
get_template_features / reset / to_device` methods, same `tracking_state` attributes, same
`{"bbox": ...}` return of `update`.  `model` is any object with `get_features(NCHW fp32)` and
`track(search, template_features)` returning the two reference dictionary keys — normally
`feartracker_amd.FEARNetHIP` (HIP kernels through the C-ABI of include/fear_hip.h).

The per-frame arithmetic follows the reference line by line where it is observable:
int32 truncation of the context box, Python banker's `round` and `max(3, .)` in the rescale,
float64 grids, first-maximum arg-max, and the optional (`smooth=True`) scale/ratio penalty,
cosine-window blend and size smoothing.
"""
from __future__ import annotations

from collections import deque
from typing import Any, Callable, Dict, Optional, Tuple, Union

import numpy as np
import torch

from .box_coder import FEARBoxCoder, TrackerDecodeResult
from .constants import TARGET_CLASSIFICATION_KEY, TARGET_REGRESSION_LABEL_KEY
from .geometry import border_color_u8, clamp_bbox, crop_geometry, get_extended_crop, normalize_image


def _resolve_device(cuda_id: Union[int, str, torch.device]) -> torch.device:
    """The reference moves tensors with `.cuda(cuda_id)` only when CUDA is available
    (utils/utils.py:11-12) and otherwise leaves them on the host."""
    if isinstance(cuda_id, torch.device):
        return cuda_id
    if isinstance(cuda_id, str):
        return torch.device(cuda_id)
    return torch.device(f"cuda:{cuda_id}") if torch.cuda.is_available() else torch.device("cpu")


def limit(radius):
    """max(r, 1/r) (reference utils/utils.py:74-77)."""
    if isinstance(radius, torch.Tensor):
        return torch.maximum(radius, 1.0 / radius)
    return np.maximum(radius, 1.0 / radius)


def squared_size(w, h):
    """sqrt((w+p)(h+p)), p = (w+h)/2 (reference utils/utils.py:80-85)."""
    pad = (w + h) * 0.5
    area = (w + pad) * (h + pad)
    return torch.sqrt(area) if isinstance(area, torch.Tensor) else np.sqrt(area)


class TrackingState:
    """Mutable per-track state; attribute names are part of the API (base_tracker.py:13-25)."""

    def __init__(self) -> None:
        self.frame_h = 0
        self.frame_w = 0
        self.bbox: Optional[np.ndarray] = None
        self.mapping: Optional[np.ndarray] = None
        self.prev_size = None
        self.mean_color = None
        self.paths: deque = deque(maxlen=10)

    def save_frame_shape(self, frame: np.ndarray) -> None:
        self.frame_h, self.frame_w = frame.shape[0], frame.shape[1]


class Tracker:
    """State, pre-processing and post-processing shared by Siamese trackers."""

    def __init__(self, model: Any, cuda_id: Union[int, str] = 0, **tracking_config: Any) -> None:
        self.cuda_id = cuda_id
        self.tracking_config = tracking_config
        self.tracking_state = TrackingState()
        self.net = model
        self.box_coder = self.get_box_coder(tracking_config, cuda_id)
        self._template_features = None
        self._template_transform = self._get_default_transform(img_size=tracking_config["template_size"])
        self._search_transform = self._get_default_transform(img_size=tracking_config["instance_size"])
        self.window = self._get_tracking_window(tracking_config["windowing"], tracking_config["score_size"])
        self.to_device(cuda_id)

    # ------------------------------------------------------------------ construction helpers
    def get_box_coder(self, tracking_config, cuda_id=0):
        raise NotImplementedError

    def to_device(self, cuda_id) -> None:
        self.cuda_id = cuda_id
        self.device = _resolve_device(cuda_id)
        self.window = self.window.to(self.device)
        self.box_coder = self.box_coder.to_device(self.device)

    @staticmethod
    def _get_tracking_window(windowing: str, score_size: int) -> torch.Tensor:
        if windowing == "cosine":
            hann = np.hanning(score_size)
            return torch.from_numpy(np.outer(hann, hann))           # float64, like the reference
        return torch.ones(int(score_size), int(score_size))

    @staticmethod
    def _get_default_transform(img_size: int) -> Callable[[np.ndarray], np.ndarray]:
        return normalize_image

    @staticmethod
    def _array_to_batch(x: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(np.transpose(x, (2, 0, 1))[None]))

    def _preprocess_image(self, image: np.ndarray, transform: Callable) -> torch.Tensor:
        """uint8 HxWxC RGB -> normalised fp32 1xCxHxW on the tracker's device (base_tracker.py:97-103)."""
        img = transform(image[:, :, :3])
        if image.shape[2] > 3:
            img = np.concatenate([img, image[:, :, 3:]], axis=2)
        return self._array_to_batch(img).float().to(self.device)

    # ------------------------------------------------------------------ geometry
    def _rescale_bbox(self, bbox: np.ndarray, padded_box) -> list:
        """Search-crop pixels -> frame pixels (base_tracker.py:83-90): separate w/h scales,
        Python `round` (half to even), minimum side 3."""
        size = self.tracking_config["instance_size"]
        sx = padded_box[2] / size
        sy = padded_box[3] / size
        bbox[0] = round(bbox[0] * sx + padded_box[0])
        bbox[1] = round(bbox[1] * sy + padded_box[1])
        bbox[2] = max(3, round(bbox[2] * sx))
        bbox[3] = max(3, round(bbox[3] * sy))
        return [int(v) for v in bbox]

    def _get_scale(self, bbox: np.ndarray) -> int:
        ctx = self.tracking_config["search_context"] * sum(bbox[2:])
        return max(round(np.sqrt((bbox[2] + ctx) * (bbox[3] + ctx))), 1)

    def _get_point_offset(self, pred_bbox: np.ndarray) -> Tuple[float, float]:
        half = self.tracking_config["instance_size"] // 2
        return pred_bbox[0] + pred_bbox[2] / 2 - half, pred_bbox[1] + pred_bbox[3] / 2 - half

    # ------------------------------------------------------------------ API stubs
    def reset(self) -> None:
        self._template_features = None

    def initialize(self, image: np.ndarray, rect: np.ndarray, **kwargs) -> None:
        pass

    def update(self, image: np.ndarray, *kw) -> Dict[str, Any]:
        return {"bbox": self.tracking_state.bbox}

    # ------------------------------------------------------------------ smooth=True branch
    def _smooth_size(self, size: np.ndarray, prev_size: np.ndarray, lr: float) -> Tuple[float, float]:
        """Size smoothing exactly as the reference computes it (base_tracker.py:126-139)."""
        size = size * lr
        prev_size = prev_size * (1 - lr)
        return (prev_size[0] + lr * (size[0] + prev_size[0]),
                prev_size[1] + lr * (size[1] + prev_size[1]))

    def _confidence_postprocess(self, cls_score: torch.Tensor, regression_map: torch.Tensor):
        """Scale/ratio penalty and window blend (base_tracker.py:166-205); identity unless `smooth`."""
        if not self.tracking_config.get("smooth", False):
            return cls_score, None
        prev = self.tracking_state.prev_size
        gx, gy = self.box_coder.grid_x, self.box_coder.grid_y
        x0 = (gx - regression_map[:, 0])[0]
        y0 = (gy - regression_map[:, 1])[0]
        x1 = (gx + regression_map[:, 2])[0]
        y1 = (gy + regression_map[:, 3])[0]
        w, h = x1 - x0, y1 - y0
        s_c = limit(squared_size(w, h) / squared_size(prev[0], prev[1]))
        r_c = limit((prev[0] / prev[1]) / (w / h))
        penalty = torch.exp(-(r_c * s_c - 1) * self.tracking_config["penalty_k"])
        influence = self.tracking_config["window_influence"]
        pscore = penalty * cls_score * (1 - influence) + self.window * influence
        return pscore, penalty.cpu().numpy()

    def _postprocess_bbox(self, decoded_info: TrackerDecodeResult, cls_score: np.ndarray, penalty: Any = None):
        pred = np.squeeze(decoded_info.bbox.cpu().numpy())
        if not self.tracking_config.get("smooth", False):
            return pred
        r, c = decoded_info.pred_coords[0]
        # the reference multiplies a numpy float64 scalar into a 0-dim fp32 torch tensor: the products are
        # rounded to fp32 (base_tracker.py:159)
        lr = float(np.float32(np.float32(penalty[r, c]) * np.float32(cls_score[r, c]))
                   * np.float32(self.tracking_config["lr"]))
        w, h = self._smooth_size(np.array(pred[2:]), prev_size=self.tracking_state.prev_size, lr=lr)
        return np.array([pred[0], pred[1], w, h])


class FEARTracker(Tracker):
    """FEAR single-object tracker front end (reference: tracker/fear_tracker.py:13-86)."""

    def get_box_coder(self, tracking_config, cuda_id: int = 0):
        return FEARBoxCoder(tracker_config=tracking_config)

    def initialize(self, image: np.ndarray, rect: np.ndarray, **kwargs) -> None:
        """image: HxWx3 uint8 RGB; rect: [x, y, w, h], 0-based."""
        rect = clamp_bbox(rect, image.shape)
        st = self.tracking_state
        st.bbox = rect
        st.paths = deque([rect], maxlen=10)
        st.mean_color = np.mean(image, axis=(0, 1))
        self._template_features = self.get_template_features(image, rect)

    def _device_crop(self, image: np.ndarray) -> bool:
        """Crop + border + resize + normalise on the GPU (`fear_crop_normalize`) whenever the model offers
        it and the frame is what that kernel reads — uint8, H x W x >= 3 — : bit-identical to the host path
        (tests/test_gpu_parity.py, tests/test_cv_parity.py: same floats, same boxes on both clips) and 7x faster per frame
        (bench.py `latency_batch1`).  Any other frame (float, uint16, grey) takes the reference-style host path, as does
        `device_crop=False` in the tracking config (not a key of the reference config; the reference always crops on the
        host with cv2, utils.py:215-253)."""
        return bool(self.tracking_config.get("device_crop", True)) and hasattr(self.net, "crop_normalize") and \
            isinstance(image, np.ndarray) and image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] >= 3

    def get_template_features(self, image: np.ndarray, rect: np.ndarray):
        cfg = self.tracking_config
        if self._device_crop(image):
            ctx, _ = crop_geometry(image.shape, rect, cfg["template_size"], cfg["template_bbox_offset"])
            pad = border_color_u8(np.mean(image, axis=(0, 1)))
            x = self.net.crop_normalize(image[:, :, :3], ctx, pad, cfg["template_size"])
            return self.net.get_features(x)
        crop, _, _ = get_extended_crop(image=image, bbox=rect, offset=cfg["template_bbox_offset"],
                                       crop_size=cfg["template_size"])
        return self.net.get_features(self._preprocess_image(crop, self._template_transform))

    def update(self, image: np.ndarray, *kw) -> Dict[str, Any]:
        cfg, st = self.tracking_config, self.tracking_state
        if self._device_crop(image):
            context, box_in_crop = crop_geometry(image.shape, st.bbox, cfg["instance_size"], cfg["search_context"])
            st.mapping = context
            st.prev_size = box_in_crop[2:]
            search = self.net.crop_normalize(image[:, :, :3], context, border_color_u8(st.mean_color), cfg["instance_size"])
            pred, _ = self._postprocess(track_result=self.net.track(search, self._template_features))
            pred = clamp_bbox(self._rescale_bbox(pred, st.mapping), image.shape)
            st.bbox = pred
            st.paths.append(pred)
            return dict(bbox=pred)
        crop, box_in_crop, context = get_extended_crop(
            image=image, bbox=st.bbox, crop_size=cfg["instance_size"], offset=cfg["search_context"],
            padding_value=st.mean_color)
        st.mapping = context
        st.prev_size = box_in_crop[2:]
        pred, _ = self.track(crop)
        pred = clamp_bbox(self._rescale_bbox(pred, st.mapping), image.shape)
        st.bbox = pred
        st.paths.append(pred)
        return dict(bbox=pred)

    def track(self, search_crop: np.ndarray):
        search = self._preprocess_image(search_crop, self._search_transform)
        return self._postprocess(track_result=self.net.track(search, self._template_features))

    def _postprocess(self, track_result: Dict[str, torch.Tensor]):
        cfg = self.tracking_config
        if cfg.get("device_postprocess", True) and hasattr(self.net, "decode_smooth") and \
                getattr(track_result.get(TARGET_CLASSIFICATION_KEY), "is_cuda", False):
            # whole post-processing in one device kernel (fear_decode / fear_decode_smooth), one 40-byte D2H; identical
            # boxes to the host path below (tests/test_gpu_parity.py); default whenever the model offers it, switched off
            # with `device_postprocess=False` (not a key of the reference config)
            cls_map, reg_map = track_result[TARGET_CLASSIFICATION_KEY], track_result[TARGET_REGRESSION_LABEL_KEY]
            if cfg.get("smooth", False):
                _, xywh, score = self.net.decode_smooth(
                    cls_map, reg_map, np.asarray(self.tracking_state.prev_size, dtype=np.float64)[None], self.window,
                    cfg["penalty_k"], cfg["window_influence"], cfg["lr"], cfg["score_size"], cfg["total_stride"],
                    cfg["instance_size"])
            else:
                _, xywh, score = self.net.decode(cls_map, reg_map, cfg["score_size"], cfg["total_stride"], cfg["instance_size"])
            if hasattr(self.net, "decoded_to_host"):          # one 44-byte transfer instead of two synchronising ones
                _, xywh_h, score_h = self.net.decoded_to_host(_, xywh, score)
                return xywh_h[0].copy(), np.float32(score_h[0])
            return xywh[0].cpu().numpy(), np.float32(score[0].item())
        reg = track_result[TARGET_REGRESSION_LABEL_KEY].detach()
        cls_score = track_result[TARGET_CLASSIFICATION_KEY].detach().float().sigmoid()
        score_map, penalty = self._confidence_postprocess(cls_score=cls_score, regression_map=reg.float())
        decoded = self.box_coder.decode(classification_map=score_map, regression_map=reg, use_sigmoid=False)
        cls_np = np.squeeze(cls_score.cpu().numpy() if isinstance(cls_score, torch.Tensor) else cls_score)
        pred = self._postprocess_bbox(decoded_info=decoded, cls_score=cls_np, penalty=penalty)
        r, c = decoded.pred_coords[0]
        return pred, cls_np[r, c]
