"""Deterministic synthetic clips for the tracker-loop parity tests (test infrastructure).

`demo_clip()` has the geometry of the reference demo (demo_video.py:44-58, assets/test.mp4): 480x256 frames,
init box [163, 53, 45, 174] — a tall object whose search context extend_bbox(., 2) = [73, -295, 225, 870] is mostly
mean-colour padding and is resized 3.4 : 1 anisotropically — moving towards and partly out of the right frame edge
while its size changes.  The frames are generated, not stored (220 x 256 x 480 x 3 bytes would be 81 MB): the fixture
tests/golden/clip_demo.npz holds their CRC32s next to the boxes the REFERENCE tracker produced on them
(tools/make_golden.py section 9), so a drifting generator is detected before any box is compared.
"""
import zlib

import numpy as np

DEMO_INIT_BBOX = (163, 53, 45, 174)        # demo_video.py:45-46
DEMO_SHAPE = (256, 480)                    # assets/test.mp4 frame size (H, W)


def demo_clip(n_frames: int = 220, seed: int = 11):
    h, w = DEMO_SHAPE
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    bg = np.stack([70 + 35 * np.sin(xx / 31.0) + 20 * np.cos(yy / 13.0),
                   90 + 30 * np.cos((xx + 2 * yy) / 41.0),
                   110 + 25 * np.sin((xx - yy) / 27.0)], axis=-1) + rng.randint(0, 16, size=(h, w, 3))
    x0, y0, bw, bh = DEMO_INIT_BBOX
    frames, boxes = [], []
    for t in range(n_frames):
        s = 1.0 - 0.25 * np.sin(np.pi * min(t, 150) / 150.0) + (0.0 if t < 150 else 0.004 * (t - 150))   # shrink, then grow
        ow, oh = bw * s, bh * s
        cx = x0 + bw / 2.0 + 1.32 * t                       # ends around x = 476: half of the object outside the frame
        cy = y0 + bh / 2.0 + 8.0 * np.sin(t / 17.0) + 0.1 * t
        inside = (np.abs(xx - cx) <= ow / 2.0) & (np.abs(yy - cy) <= oh / 2.0)
        u, v = (xx - cx) / ow, (yy - cy) / oh              # object-fixed coordinates: the texture moves and scales with it
        tex = np.stack([215 + 35 * np.sin(u * 19.0), 60 + 50 * np.cos(v * 23.0), 150 + 70 * np.sin((u + v) * 13.0)], axis=-1)
        # a darker "head" ellipse so the object is not symmetric
        head = ((u / 0.35) ** 2 + ((v + 0.32) / 0.12) ** 2) <= 1.0
        tex[head] = tex[head] * 0.35
        f = bg.copy()
        f[inside] = tex[inside]
        frames.append(np.clip(f, 0, 255).astype(np.uint8))
        boxes.append([int(cx - ow / 2.0), int(cy - oh / 2.0), int(ow), int(oh)])
    return np.stack(frames), np.array(boxes)


def frame_crcs(frames: np.ndarray) -> np.ndarray:
    return np.array([zlib.crc32(np.ascontiguousarray(f).tobytes()) for f in frames], dtype=np.uint32)
