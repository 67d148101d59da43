"""Dictionary keys of the network output, part of the drop-in API surface
(reference: model_training/utils/constants.py:1,3)."""

TARGET_CLASSIFICATION_KEY = "TARGET_CLASSIFICATION_KEY"
TARGET_REGRESSION_LABEL_KEY = "TARGET_REGRESSION_LABEL_KEY"

# shipped tracker configuration, model_training/config/tracker/siam_tracker.yaml:1-15
# (`stride` resolves to model.stride = 2, model/fear.yaml:12).  `smooth` is deliberately absent,
# exactly like the reference YAML, so the penalty/window branch is off by default.
DEFAULT_TRACKING_CONFIG = dict(
    penalty_k=0.062,
    window_influence=0.38,
    lr=0.765,
    windowing="cosine",
    total_stride=16,
    score_size=16,
    ratio=0.94,
    stride=2,
    bbox_ratio=0.5,
    template_bbox_offset=0.2,
    search_context=2,
    instance_size=256,
    template_size=128,
)
