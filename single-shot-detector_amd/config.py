"""The reference's JSON config surface (config_mobilenet.json / config_shufflenet.json).

Only the keys the inference graph consumes are used (model.py:22-30,46,57-61;
create_pb.py:24); training keys are accepted and ignored so the reference's files load
unmodified.
"""
import json

INFERENCE_KEYS = ("backbone", "depth_multiplier", "num_classes", "score_threshold",
                  "iou_threshold", "max_boxes_per_class", "min_dimension")

_DEFAULTS = {"min_dimension": 640}      # create_pb.py:24 MIN_DIMENSION


def load_config(path_or_dict):
    """Returns a dict with exactly INFERENCE_KEYS.  Raises KeyError / ValueError like the
    reference would fail on a malformed config (model.py indexes params[...] directly)."""
    if isinstance(path_or_dict, dict):
        raw = dict(path_or_dict)
    else:
        with open(path_or_dict) as f:
            raw = json.load(f)
    out = {}
    for k in INFERENCE_KEYS:
        if k in raw:
            out[k] = raw[k]
        elif k in _DEFAULTS:
            out[k] = _DEFAULTS[k]
        else:
            raise KeyError("config is missing the key %r" % k)
    if out["backbone"] not in ("mobilenet", "shufflenet"):
        raise ValueError("backbone must be 'mobilenet' or 'shufflenet' (model.py:22-30)")
    out["depth_multiplier"] = float(out["depth_multiplier"])
    out["num_classes"] = int(out["num_classes"])
    out["score_threshold"] = float(out["score_threshold"])
    out["iou_threshold"] = float(out["iou_threshold"])
    out["max_boxes_per_class"] = int(out["max_boxes_per_class"])
    out["min_dimension"] = int(out["min_dimension"])
    return out
