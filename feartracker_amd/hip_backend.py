"""ctypes binding of the C ABI in include/fear_hip.h and the drop-in model object.

`FEARNetHIP` exposes the surface of the reference `FEARNet` that the tracker, the CoreML
wrapper and the thop wrapper use (model_training/model/fear_net.py:58-96,
evaluate/coreml_convert.py:55-57, evaluate/macs_params.py:15-17):
`get_features(crop)`, `track(search, template_features)`, `forward((template, search))`,
`connector(...)` is internal to the fused engine, `.eval()/.cuda()/.to()` are no-ops returning
self.  All tensors are fp32 NCHW torch tensors on the handle's GPU; torch is only the
allocator/stream provider here — every FLOP runs in the hand-written HIP kernels.

There is NO fallback: if the shared library is missing or the GPU is absent the constructor
raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch  # must be imported before the library so that both share one libamdhip64

from .constants import TARGET_CLASSIFICATION_KEY, TARGET_REGRESSION_LABEL_KEY

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FEAR_LIB", os.path.join(_PKG, "libfear_hip.so"))   # FEAR_LIB: development builds (tools/)
DEFAULT_WEIGHTS = os.path.join(_PKG, "weights", "fear_xs_noembs.fearw")
WEIGHTS_FEAR_M = os.path.join(_PKG, "weights", "fear_m_synth.fearw")   # synthetic deeper trunk (tools/make_fear_m.py), random weights

FEAR_OPT_MAX_BATCH = 1
FEAR_OPT_PROFILE = 2
FEAR_OPT_PROFILE_OP = 3
FEAR_OPT_FUSE = 4
FEAR_OPT_MATH = 5
FEAR_OPT_CHAIN = 6
FEAR_OPT_SMALL_PASS = 7
FEAR_OPT_PLAN_CROPS = 8
FEAR_OPT_DUAL_HEAD = 9
FEAR_OPT_HEAD_STAGGER = 10
FEAR_OPT_TILE_V4 = 11
FEAR_OPT_TINY_SEP = 12
FEAR_OPT_HEAD_CHAIN = 13
FEAR_OPT_BF16_STORE = 14
FEAR_OPT_E1_PAIR = 15
FEAR_OPT_SPLIT_STREAMS = 16
FEAR_OPT_CHAIN32 = 17

_lib = None


def load_library() -> ctypes.CDLL:
    """dlopen libfear_hip.so and declare the prototypes of include/fear_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found. Build it with `python -m feartracker_amd._build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, i32, i64, f32p = c.c_void_p, c.c_int, c.c_int64, c.c_void_p
    lib.fear_create.argtypes = [vp, c.c_size_t, i32, c.POINTER(vp)]
    lib.fear_create.restype = i32
    lib.fear_destroy.argtypes = [vp]
    lib.fear_destroy.restype = i32
    lib.fear_features.argtypes = [vp, f32p, i32, i32, f32p, vp]
    lib.fear_features.restype = i32
    lib.fear_track.argtypes = [vp, f32p, f32p, f32p, i32, f32p, f32p, vp]
    lib.fear_track.restype = i32
    lib.fear_track_packed.argtypes = [vp, f32p, f32p, f32p, i32, f32p, vp]
    lib.fear_track_packed.restype = i32
    lib.fear_decode.argtypes = [vp, f32p, f32p, i32, i32, i32, i32, vp, vp, f32p, vp]
    lib.fear_decode.restype = i32
    f64 = ctypes.c_double
    lib.fear_decode_smooth.argtypes = [vp, f32p, f32p, i32, i32, i32, i32, vp, vp, f64, f64, f64, vp, vp, f32p, vp]
    lib.fear_decode_smooth.restype = i32
    lib.fear_normalize_u8.argtypes = [vp, vp, i32, i32, f32p, vp]
    lib.fear_normalize_u8.restype = i32
    lib.fear_crop_normalize.argtypes = [vp, vp, i32, i32, vp, vp, i32, i32, f32p, vp]
    lib.fear_crop_normalize.restype = i32
    lib.fear_set_option.argtypes = [vp, i32, i64]
    lib.fear_set_option.restype = i32
    lib.fear_get_option.argtypes = [vp, i32]
    lib.fear_get_option.restype = i64
    lib.fear_plan_size.argtypes = [vp, i32, i32]
    lib.fear_plan_size.restype = i32
    lib.fear_plan_op.argtypes = [vp, i32, i32, i32, c.c_char_p, c.POINTER(c.c_double), c.POINTER(c.c_double)]
    lib.fear_plan_op.restype = i32
    lib.fear_profile_read.argtypes = [vp, i32, i32, i32, c.POINTER(c.c_double), c.POINTER(i64)]
    lib.fear_profile_read.restype = i32
    lib.fear_profile_reset.argtypes = [vp]
    lib.fear_profile_reset.restype = i32
    lib.fear_workspace_bytes.argtypes = [vp]
    lib.fear_workspace_bytes.restype = c.c_size_t
    lib.fear_strerror.argtypes = [i32]
    lib.fear_strerror.restype = c.c_char_p
    lib.fear_last_hip_error.argtypes = [vp]
    lib.fear_last_hip_error.restype = i32
    lib.fear_version.argtypes = []
    lib.fear_version.restype = c.c_char_p
    _lib = lib
    return lib


EXPORTED_SYMBOLS = (
    "fear_create", "fear_destroy", "fear_features", "fear_track", "fear_track_packed", "fear_decode", "fear_decode_smooth", "fear_normalize_u8",
    "fear_crop_normalize",
    "fear_set_option", "fear_get_option", "fear_plan_size", "fear_plan_op", "fear_profile_read",
    "fear_profile_reset", "fear_workspace_bytes", "fear_strerror", "fear_last_hip_error", "fear_version",
)


class FearError(RuntimeError):
    pass


def context_rectangle(frame_h: int, frame_w: int, ctx_xywh: np.ndarray) -> Tuple[int, int, int, int]:
    """(x0, y0, x1, y1): the part of an H x W frame that context boxes (n, 4) int xywh can sample — their union clipped to the
    frame; when nothing of the frame is visible one pixel keeps the shapes legal (every sample is border colour then anyway)."""
    ctx = np.asarray(ctx_xywh).reshape(-1, 4)
    x0 = y0 = x1 = y1 = 0
    if ctx.shape[0] == 1:                                   # the tracker's case: plain integers
        cx, cy, cw, ch = (int(v) for v in ctx[0])
        x0, y0 = min(max(cx, 0), frame_w), min(max(cy, 0), frame_h)
        x1, y1 = min(max(cx + cw, x0), frame_w), min(max(cy + ch, y0), frame_h)
    elif ctx.shape[0]:
        ctx = ctx.astype(np.int64)
        x0 = int(np.clip(ctx[:, 0].min(), 0, frame_w))
        y0 = int(np.clip(ctx[:, 1].min(), 0, frame_h))
        x1 = int(np.clip((ctx[:, 0] + ctx[:, 2]).max(), x0, frame_w))
        y1 = int(np.clip((ctx[:, 1] + ctx[:, 3]).max(), y0, frame_h))
    if x1 <= x0 or y1 <= y0:
        x0, y0 = min(x0, frame_w - 1), min(y0, frame_h - 1)
        x1, y1 = x0 + 1, y0 + 1
    return x0, y0, x1, y1


class FEARNetHIP:
    """FEAR network running on one MI355X through libfear_hip.so.

    One handle = one device, one workspace: calls are ordered on the device even when they are issued on different torch
    streams (the engine makes a call on a new stream wait for the previous call's work); the object is not thread-safe."""

    def __init__(self, weights_path: str = DEFAULT_WEIGHTS, device: int = 0, max_batch: int = 64):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        if not torch.cuda.is_available():
            raise RuntimeError("FEARNetHIP needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.device = torch.device(f"cuda:{int(device)}")
        torch.cuda.init()
        with torch.cuda.device(self.device):
            torch.zeros(1, device=self.device)  # make sure the HIP context exists before the engine uploads
            with open(weights_path, "rb") as fh:
                blob = fh.read()
            self._check(self._lib.fear_create(blob, len(blob), int(device), ctypes.byref(self._h)))
        self.weights_path = weights_path
        self.set_max_batch(max_batch)
        self.feat_channels = 256

    # ------------------------------------------------------------------ plumbing
    def _check(self, status: int) -> None:
        if status != 0:
            hip = self._lib.fear_last_hip_error(self._h) if self._h else 0
            raise FearError(f"libfear_hip: {self._lib.fear_strerror(status).decode()} (status {status}, hip {hip})")

    def _stream(self) -> ctypes.c_void_p:
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _prep(self, t: torch.Tensor, name: str) -> torch.Tensor:
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name} must be a torch.Tensor")
        if t.device != self.device:
            t = t.to(self.device)
        if t.dtype != torch.float32:
            t = t.float()
        return t.contiguous()

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h:
                self._lib.fear_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ nn.Module look-alikes
    def eval(self):
        return self

    def cuda(self, device=None):
        return self

    def to(self, *a, **k):
        return self

    def set_max_batch(self, n: int) -> None:
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_MAX_BATCH, int(n)))

    def set_fuse(self, on: bool) -> None:
        """Fused block kernels (default) vs one kernel per conv layer (bring-up / A-B measurements)."""
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_FUSE, 1 if on else 0))

    def set_chain(self, on: bool) -> None:
        """Stride-16 trunk stage + neck as one chain kernel (default, fp32 mode) vs one fused kernel per block."""
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_CHAIN, 1 if on else 0))

    def set_small_pass(self, crops: int) -> None:
        """Passes of at most `crops` crops run the small-batch plan (split-K 16x16 kernels, two-stream head); 0 = never."""
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_SMALL_PASS, int(crops)))

    def set_dual_head(self, on: bool) -> None:
        """Throughput plan: the head's two branches on two streams (default) vs one."""
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_DUAL_HEAD, 1 if on else 0))

    def set_head_chain(self, on: bool) -> None:
        """A/B switch for the one-launch BoxTower (FEAR_OPT_HEAD_CHAIN, default on; throughput plan): off = the eight sep16
        launches it replaces.  fp32 mode (set_math(0)): headchain_kernel, maps bit-identical either way.  bf16 mode (set_math(2)):
        headchain_b_kernel, same rounding points as the launches but another summation order — NOT bit-identical (2e-3 relative on
        the maps; tests/test_gpu_parity.py).  set_math(1) keeps the launches.  While the chain is on, set_dual_head has no effect
        on the throughput plan (there are no separate branch launches to put on two streams)."""
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_HEAD_CHAIN, 1 if on else 0))

    def set_bf16_store(self, on: bool) -> None:
        """A/B switch (FEAR_OPT_BF16_STORE, default on; only with set_math(2)): bf16 storage of the activations between the kernels of
        the trunk's HBM-bound front."""
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_BF16_STORE, 1 if on else 0))

    def set_e1_pair(self, on: bool) -> None:
        """A/B switch (FEAR_OPT_E1_PAIR, default on; fp32 mode, throughput plan): two consecutive 24-channel e1 blocks as one launch
        (the map between them stays in LDS) vs one tile-kernel launch per block."""
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_E1_PAIR, 1 if on else 0))

    def set_chain32(self, mode) -> None:
        """A/B switch (FEAR_OPT_CHAIN32; fp32 mode, throughput plan): the four blocks of the 32 x 32 trunk stage as a register-resident
        chain — 2 (default, also True): in one launch with the stride-16 stage + neck (chain32_16_kernel), 1: a launch of its own
        (chain32_kernel), 0 / False: one tile-kernel launch per block."""
        mode = 2 if mode is True else int(mode)
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_CHAIN32, mode))

    def set_split_streams(self, on: bool) -> None:
        """FEAR_OPT_SPLIT_STREAMS (default off): a throughput pass of `track` / `track_maps` runs as two half-batches on two HIP
        streams of the same handle — a serving option (+3 % crops/s at 256 crops), bit-identical maps; bench.py keeps it off for
        `value` so that its per-kernel figures stay full-grid launches and prints its number beside it."""
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_SPLIT_STREAMS, 1 if on else 0))

    def set_tile_v4(self, on: bool) -> None:
        """Throughput plan: the phase-overlapped tile kernel for the blocks that have one (default on) vs ir_tile_v2 everywhere."""
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_TILE_V4, 1 if on else 0))


    def set_tiny_sep(self, on: bool) -> None:
        """A/B switch for the tiny plan's row-split SepConv slice kernel (FEAR_OPT_TINY_SEP, default on)."""
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_TINY_SEP, 1 if on else 0))
    def set_head_stagger(self, microseconds: int) -> None:
        """Two head streams: hold the second branch back by this many microseconds (FEAR_OPT_HEAD_STAGGER)."""
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_HEAD_STAGGER, int(microseconds)))

    def set_plan_crops(self, crops: int) -> None:
        """Crop count whose launch plan `plan()` / `profile_read()` describe (0 = a full pass of max_batch crops)."""
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_PLAN_CROPS, int(crops)))

    def set_math(self, mode: int) -> None:
        """0: exact fp32 MFMA (default); 1: fp16 hi+lo split operands on the matrix pipe, fp32 accumulate (fp32-grade);
        2: bf16 operands, fp32 accumulate (reduced precision: the bf16 MFMA pointwise-conv path of BASELINE configs[3])."""
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_MATH, int(mode)))

    def set_profile(self, on: bool, op: int = -1) -> None:
        """Bracket kernel launches with hipEvents; op >= 0 restricts it to one op of the plan."""
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_PROFILE_OP, int(op)))
        self._check(self._lib.fear_set_option(self._h, FEAR_OPT_PROFILE, 1 if on else 0))

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def get_features(self, crop: torch.Tensor) -> torch.Tensor:
        """(N,3,H,H) normalised fp32 -> (N,256,H/16,H/16); fear_net.py:63-66."""
        crop = self._prep(crop, "crop")
        if crop.dim() != 4 or crop.shape[1] != 3 or crop.shape[2] != crop.shape[3]:
            raise ValueError(f"crop must be (N,3,H,H), got {tuple(crop.shape)}")
        n, hw = crop.shape[0], crop.shape[2]
        out = torch.empty((n, self.feat_channels, hw // 16, hw // 16), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self._lib.fear_features(self._h, crop.data_ptr(), n, hw, out.data_ptr(), self._stream()))
        return out

    @torch.no_grad()
    def track(self, search: torch.Tensor, template_features: torch.Tensor,
              update: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """fear_net.py:90-96; `update` = optional cls-branch template (blocks.py:174-179)."""
        bbox, cls = self.track_maps(search, template_features, update)
        return {TARGET_REGRESSION_LABEL_KEY: bbox, TARGET_CLASSIFICATION_KEY: cls}

    def _track_inputs(self, search, template_features, update):
        """Shared argument checks of `track_maps` / `track_packed`: (search, z, pointer of the optional cls template, n).
        A single template (N = 1) is broadcast over the batch, as the reference's `expand` does."""
        search = self._prep(search, "search")
        z = self._prep(template_features, "template_features")
        n = search.shape[0]
        if tuple(search.shape[1:]) != (3, 256, 256):
            raise ValueError(f"search must be (N,3,256,256), got {tuple(search.shape)}")
        if z.shape[0] == 1 and n > 1:
            z = z.expand(n, -1, -1, -1).contiguous()
        if tuple(z.shape) != (n, self.feat_channels, 8, 8):
            raise ValueError(f"template_features must be ({n},256,8,8), got {tuple(z.shape)}")
        zu = None
        if update is not None:
            zu = self._prep(update, "update")
            if zu.shape[0] == 1 and n > 1:
                zu = zu.expand(n, -1, -1, -1).contiguous()
            if tuple(zu.shape) != tuple(z.shape):
                raise ValueError("update template must have the shape of template_features")
        return search, z, zu, n

    @torch.no_grad()
    def track_maps(self, search, template_features, update=None, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        search, z, zu, n = self._track_inputs(search, template_features, update)
        if out is None:
            bbox = torch.empty((n, 4, 16, 16), dtype=torch.float32, device=self.device)
            cls = torch.empty((n, 1, 16, 16), dtype=torch.float32, device=self.device)
        else:
            bbox, cls = out
        with torch.cuda.device(self.device):
            self._check(self._lib.fear_track(self._h, search.data_ptr(), z.data_ptr(), zu.data_ptr() if zu is not None else None, n,
                                             bbox.data_ptr(), cls.data_ptr(), self._stream()))
        return bbox, cls

    @torch.no_grad()
    def track_packed(self, search, template_features, update=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`track` with both maps written into one (N,5,16,16) tensor — bbox in channels 0..3, cls in channel 4 —
        the payload of the multi-GPU all-gather (sharding.py); no pack/cat kernel."""
        search, z, zu, n = self._track_inputs(search, template_features, update)
        if out is None:
            out = torch.empty((n, 5, 16, 16), dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != (n, 5, 16, 16) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous fp32 (N,5,16,16) tensor on the engine's device")
        with torch.cuda.device(self.device):
            self._check(self._lib.fear_track_packed(self._h, search.data_ptr(), z.data_ptr(), zu.data_ptr() if zu is not None else None,
                                                    n, out.data_ptr(), self._stream()))
        return out

    @torch.no_grad()
    def forward(self, x: Tuple[torch.Tensor, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """fear_net.py:83-88: both branches through the trunk."""
        template, search = x
        return self.track(search, self.get_features(template))

    __call__ = forward

    # ------------------------------------------------------------------ device-side helpers
    def _decode_outputs(self, n: int):
        """The three outputs of the decode kernels as views of ONE device buffer [xywh f64 (n,4) | rc i32 (n,2) | score f32 (n)],
        so that a caller who wants them on the host fetches 44 bytes per crop in one transfer (`decoded_to_host`)."""
        buf = torch.empty(n * 44 + 4, dtype=torch.uint8, device=self.device)
        xywh = buf[: n * 32].view(torch.float64).view(n, 4)
        rc = buf[n * 32: n * 40].view(torch.int32).view(n, 2)
        score = buf[n * 40: n * 44].view(torch.float32)
        return buf, rc, xywh, score

    @staticmethod
    def decoded_to_host(rc: torch.Tensor, xywh: torch.Tensor, score: torch.Tensor):
        """(rc, xywh, score) of `decode` / `decode_smooth` as numpy arrays with ONE device-to-host copy (they are views of one
        buffer; three `.cpu()` calls would be three synchronising transfers)."""
        n = xywh.shape[0]
        # only the exact views `_decode_outputs` makes share one 44-byte-per-crop buffer: check the layout instead of assuming it
        base = xywh.storage_offset() * 8
        st = xywh.untyped_storage()
        ok = (xywh.dtype == torch.float64 and rc.dtype == torch.int32 and score.dtype == torch.float32 and
              tuple(xywh.shape) == (n, 4) and tuple(rc.shape) == (n, 2) and tuple(score.shape) == (n,) and
              xywh.is_contiguous() and rc.is_contiguous() and score.is_contiguous() and
              rc.untyped_storage().data_ptr() == st.data_ptr() and score.untyped_storage().data_ptr() == st.data_ptr() and
              rc.storage_offset() * 4 == base + n * 32 and score.storage_offset() * 4 == base + n * 40 and
              st.nbytes() >= base + n * 44)
        if not ok:              # any other tensors (slices, another backend's outputs): three plain copies
            return rc.cpu().numpy(), xywh.cpu().numpy(), score.cpu().numpy()
        host = torch.empty(0, dtype=torch.uint8, device=xywh.device).set_(st, base, (n * 44,)).cpu().numpy()
        return (host[n * 32: n * 40].view(np.int32).reshape(n, 2), host[: n * 32].view(np.float64).reshape(n, 4),
                host[n * 40: n * 44].view(np.float32))

    @torch.no_grad()
    def decode(self, cls: torch.Tensor, bbox: torch.Tensor, score_size: int = 16, total_stride: int = 16,
               instance_size: int = 256):
        """Device arg-max decode (box_coder.py:75-107 with use_sigmoid=True).
        Returns (rc int32 (N,2), xywh float64 (N,4), score fp32 (N,))."""
        cls = self._prep(cls, "cls")
        bbox = self._prep(bbox, "bbox")
        n = cls.shape[0]
        _, rc, xywh, score = self._decode_outputs(n)
        with torch.cuda.device(self.device):
            self._check(self._lib.fear_decode(self._h, cls.data_ptr(), bbox.data_ptr(), n, score_size, total_stride,
                                              instance_size, rc.data_ptr(), xywh.data_ptr(), score.data_ptr(),
                                              self._stream()))
        return rc, xywh, score

    @torch.no_grad()
    def decode_smooth(self, cls: torch.Tensor, bbox: torch.Tensor, prev_size, window, penalty_k: float,
                      window_influence: float, lr: float, score_size: int = 16, total_stride: int = 16,
                      instance_size: int = 256):
        """Device `smooth=True` post-processing (base_tracker.py:149-205), batched: penalty against `prev_size`
        (N,2), window blend, arg-max, decode, size smoothing.  Returns (rc int32 (N,2), xywh float64 (N,4), score (N,))."""
        cls = self._prep(cls, "cls")
        bbox = self._prep(bbox, "bbox")
        n = cls.shape[0]
        prev = torch.as_tensor(prev_size, dtype=torch.float64).reshape(-1, 2).to(self.device).contiguous()
        win = torch.as_tensor(window, dtype=torch.float64).reshape(-1).to(self.device).contiguous()
        if prev.shape[0] != n or win.numel() != score_size * score_size:
            raise ValueError("prev_size must be (N,2) and window (score_size, score_size)")
        _, rc, xywh, score = self._decode_outputs(n)
        with torch.cuda.device(self.device):
            self._check(self._lib.fear_decode_smooth(self._h, cls.data_ptr(), bbox.data_ptr(), n, score_size, total_stride,
                                                     instance_size, prev.data_ptr(), win.data_ptr(), float(penalty_k),
                                                     float(window_influence), float(lr), rc.data_ptr(), xywh.data_ptr(),
                                                     score.data_ptr(), self._stream()))
        return rc, xywh, score

    @torch.no_grad()
    def normalize_u8(self, crops_u8_nhwc: torch.Tensor) -> torch.Tensor:
        """uint8 (N,H,H,3) RGB -> normalised fp32 (N,3,H,H) on device (base_tracker.py:97-103)."""
        x = crops_u8_nhwc
        if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[3] != 3 or x.shape[1] != x.shape[2]:
            raise ValueError("expected uint8 (N,H,H,3)")
        x = x.to(self.device).contiguous()
        n, hw = x.shape[0], x.shape[1]
        out = torch.empty((n, 3, hw, hw), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self._lib.fear_normalize_u8(self._h, x.data_ptr(), n, hw, out.data_ptr(), self._stream()))
        return out

    @torch.no_grad()
    def crop_normalize(self, frame_u8, ctx_xywh, pad_rgb_u8, out_hw: int) -> torch.Tensor:
        """Device get_extended_crop + normalise (utils.py:215-253 + base_tracker.py:97-103): frame (H,W,3) uint8 — a numpy
        array / CPU tensor (uploaded here) or a tensor already on the GPU —, ctx_xywh (n,4) int context boxes, pad_rgb_u8
        (n,3) uint8 border colours -> (n,3,out_hw,out_hw) fp32.

        Of a HOST frame only the rectangle the context boxes can sample travels over PCIe (their union clipped to the frame:
        a 225x870 context in a 1080p frame is 0.3 MB of the frame's 6.2 MB; the boxes are shifted to that rectangle's
        origin, everything outside it is border colour for the kernel exactly as the rest of the frame's outside is), and
        the context boxes and border colours go up in ONE small transfer."""
        is_np = isinstance(frame_u8, np.ndarray)          # (kept as numpy until the rectangle is cut: views with negative strides —
        if is_np:                                         #  a BGR -> RGB flip `img[:, :, ::-1]` — are fine for numpy, not for from_numpy)
            if frame_u8.dtype != np.uint8 or frame_u8.ndim != 3 or frame_u8.shape[2] != 3:
                raise ValueError("frame must be uint8 (H,W,3)")
        elif frame_u8.dtype != torch.uint8 or frame_u8.dim() != 3 or frame_u8.shape[2] != 3:
            raise ValueError("frame must be uint8 (H,W,3)")
        ctx_np = np.ascontiguousarray(np.asarray(ctx_xywh, dtype=np.int32).reshape(-1, 4))
        pad_np = np.ascontiguousarray(np.asarray(pad_rgb_u8, dtype=np.uint8).reshape(-1, 3))
        n = ctx_np.shape[0]
        if pad_np.shape[0] != n:
            raise ValueError("one border colour per context box")
        meta_bytes = (n * 16 + n * 3 + 15) // 16 * 16                       # [n x 4 int32 boxes | n x 3 uint8 colours], 16-byte padded
        if is_np or not frame_u8.is_cuda:
            # ONE transfer: [meta | the context rectangle of the frame] assembled in one host buffer (a reused pinned staging
            # buffer was tried — ADVICE r1 — and measured 10x SLOWER per frame on the 256-core host: torch's CPU->pinned copy_
            # costs milliseconds there; a plain pageable .to() of a fresh small array does not)
            fh, fw = int(frame_u8.shape[0]), int(frame_u8.shape[1])
            x0, y0, x1, y1 = context_rectangle(fh, fw, ctx_np)
            if (x1 - x0, y1 - y0) != (fw, fh):
                ctx_np = ctx_np.copy()
                ctx_np[:, 0] -= x0
                ctx_np[:, 1] -= y0
            rh, rw = y1 - y0, x1 - x0
            host = np.empty(meta_bytes + rh * rw * 3, dtype=np.uint8)
            host[: n * 16] = ctx_np.view(np.uint8).reshape(-1)
            host[n * 16: n * 16 + n * 3] = pad_np.reshape(-1)
            roi = frame_u8[y0:y1, x0:x1]
            host[meta_bytes:].reshape(rh, rw, 3)[...] = roi if is_np else roi.numpy()    # (numpy copies any strides, also negative ones)
            dev = torch.from_numpy(host).to(self.device)
            meta_ptr, frame_ptr = dev.data_ptr(), dev.data_ptr() + meta_bytes
        else:
            frame_u8 = frame_u8.to(self.device).contiguous()
            rh, rw = int(frame_u8.shape[0]), int(frame_u8.shape[1])
            meta = np.zeros(meta_bytes, dtype=np.uint8)
            meta[: n * 16] = ctx_np.view(np.uint8).reshape(-1)
            meta[n * 16: n * 16 + n * 3] = pad_np.reshape(-1)
            dev = torch.from_numpy(meta).to(self.device)
            meta_ptr, frame_ptr = dev.data_ptr(), frame_u8.data_ptr()
        out = torch.empty((n, 3, out_hw, out_hw), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self._lib.fear_crop_normalize(self._h, frame_ptr, rh, rw, meta_ptr, meta_ptr + n * 16, n, int(out_hw),
                                                      out.data_ptr(), self._stream()))
        return out

    # ------------------------------------------------------------------ measurement
    def plan(self, hw: int = 256, with_head: bool = True):
        """[(name, flops_per_crop, bytes_per_crop)] for every kernel launch of the plan."""
        n = self._lib.fear_plan_size(self._h, hw, int(with_head))
        if n < 0:
            self._check(n)
        ops = []
        for i in range(n):
            name = ctypes.create_string_buffer(64)
            fl, by = ctypes.c_double(), ctypes.c_double()
            self._check(self._lib.fear_plan_op(self._h, hw, int(with_head), i, name, ctypes.byref(fl), ctypes.byref(by)))
            ops.append((name.value.decode(), fl.value, by.value))
        return ops

    def profile_read(self, hw: int = 256, with_head: bool = True):
        """[(total_ms, launches)] per op since the last reset (needs set_profile(True))."""
        n = self._lib.fear_plan_size(self._h, hw, int(with_head))
        res = []
        for i in range(n):
            ms, cnt = ctypes.c_double(), ctypes.c_int64()
            self._check(self._lib.fear_profile_read(self._h, hw, int(with_head), i, ctypes.byref(ms), ctypes.byref(cnt)))
            res.append((ms.value, cnt.value))
        return res

    def profile_reset(self) -> None:
        self._check(self._lib.fear_profile_reset(self._h))

    def workspace_bytes(self) -> int:
        return int(self._lib.fear_workspace_bytes(self._h))
