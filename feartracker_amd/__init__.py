"""feartracker_amd — MI355X-native FEAR per-frame inference path.

Hot path only: `FEARTracker.initialize()/update()` on top of
`FEARNetHIP.get_features()/track()`, whose arithmetic runs in hand-written gfx950 HIP kernels
behind the C ABI of include/fear_hip.h.
"""
from .constants import DEFAULT_TRACKING_CONFIG, TARGET_CLASSIFICATION_KEY, TARGET_REGRESSION_LABEL_KEY
from .box_coder import FEARBoxCoder, TrackerDecodeResult, TrackerEncodeResult
from .tracker import FEARTracker, Tracker, TrackingState
from .hip_backend import FEARNetHIP, FearError, load_library, DEFAULT_WEIGHTS, LIB_PATH

__all__ = [
    "DEFAULT_TRACKING_CONFIG", "TARGET_CLASSIFICATION_KEY", "TARGET_REGRESSION_LABEL_KEY",
    "FEARBoxCoder", "TrackerDecodeResult", "TrackerEncodeResult", "FEARTracker", "Tracker", "TrackingState",
    "FEARNetHIP", "FearError", "load_library", "DEFAULT_WEIGHTS", "LIB_PATH",
]
