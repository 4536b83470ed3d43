"""CPU ORACLE -- the reference's inference graph restated on top of oracle/ops.py.

TEST INFRASTRUCTURE ONLY (see ssd_oracle.c).  PARITY UNPINNED against TensorFlow.
`W` is a dict {TF variable name: float32 ndarray} with the reference's variable names and
shapes (SURVEY.md section 8a "Weights").  All tensors NHWC float32.
"""
import numpy as np

from . import ops


def _bn(x, W, scope, act):
    return ops.bn_act(x, W[scope + "/gamma"], W[scope + "/beta"], W[scope + "/moving_mean"],
                      W[scope + "/moving_variance"], act)


# ---------------------------------------------------------------- mobilenet_v1.py:7-73
MOBILENET_LAYERS = [(1, 64), (2, 128), (1, 128), (2, 256), (1, 256), (2, 512), (1, 512),
                    (1, 512), (1, 512), (1, 512), (1, 512), (2, 1024), (1, 1024)]


def mobilenet_v1(x, W, features=None):
    """x: already standardised image (2*img - 1), [B,H,W,3].  Returns c3, c4, c5."""
    feats = {} if features is None else features
    s = "MobilenetV1/Conv2d_0"
    x = _bn(ops.conv2d(x, W[s + "/weights"], stride=2, mode="SAME"), W, s + "/BatchNorm", "relu6")
    feats["Conv2d_0"] = x
    for i, (stride, _f) in enumerate(MOBILENET_LAYERS, 1):
        s = "MobilenetV1/Conv2d_%d_depthwise" % i           # depthwise_conv.py:5-26
        x = _bn(ops.depthwise3x3(x, W[s + "/depthwise_weights"], stride), W, s + "/BatchNorm",
                "relu6")
        feats["Conv2d_%d_depthwise" % i] = x
        s = "MobilenetV1/Conv2d_%d_pointwise" % i
        x = _bn(ops.conv2d(x, W[s + "/weights"], 1, "SAME"), W, s + "/BatchNorm", "relu6")
        feats["Conv2d_%d_pointwise" % i] = x
    return {"c3": feats["Conv2d_5_pointwise"], "c4": feats["Conv2d_11_pointwise"],
            "c5": feats["Conv2d_13_pointwise"]}


# ---------------------------------------------------------------- shufflenet_v2.py:7-137
def _sn_conv(x, W, scope, act="relu"):
    return _bn(ops.conv2d(x, W[scope + "/weights"], 1, "SAME"), W, scope + "/batch_norm", act)


def _sn_dw(x, W, scope, stride):
    # depthwise_conv(..., activation_fn=None): BN, no activation (shufflenet_v2.py:121,131,135)
    return _bn(ops.depthwise3x3(x, W[scope + "/depthwise_weights"], stride), W,
               scope + "/batch_norm", None)


def _basic_unit(x, W, scope):                                # :118-123
    x = _sn_conv(x, W, scope + "/conv1x1_before")
    x = _sn_dw(x, W, scope + "/depthwise", 1)
    return _sn_conv(x, W, scope + "/conv1x1_after")


def _basic_unit_with_downsampling(x, W, scope):              # :126-137
    y = _sn_conv(x, W, scope + "/conv1x1_before")
    y = _sn_dw(y, W, scope + "/depthwise", 2)
    y = _sn_conv(y, W, scope + "/conv1x1_after")
    x = _sn_dw(x, W, scope + "/second_branch/depthwise", 2)
    x = _sn_conv(x, W, scope + "/second_branch/conv1x1_after")
    return x, y


def _block(x, W, scope, num_units):                          # :79-91
    x, y = _basic_unit_with_downsampling(x, W, scope + "/unit_1")
    for j in range(2, num_units + 1):
        x, y = ops.concat_shuffle_split(x, y)
        x = _basic_unit(x, W, scope + "/unit_%d" % j)
    return np.concatenate([x, y], axis=3)


def shufflenet_v2(x, W, features=None):
    feats = {} if features is None else features
    s = "ShuffleNetV2/Conv1"
    x = _bn(ops.conv2d(x, W[s + "/weights"], 2, "SAME"), W, s + "/batch_norm", "relu")
    feats["Conv1"] = x
    x = ops.maxpool3x3s2(x)
    feats["MaxPool"] = x
    x = _block(x, W, "ShuffleNetV2/Stage2", 4)
    feats["Stage2"] = x
    x = _block(x, W, "ShuffleNetV2/Stage3", 8)
    feats["Stage3"] = x
    x = _block(x, W, "ShuffleNetV2/Stage4", 4)
    feats["Stage4"] = x
    x = _sn_conv(x, W, "ShuffleNetV2/Conv5")
    feats["Conv5"] = x
    return {"c3": feats["Stage2"], "c4": feats["Stage3"], "c5": feats["Conv5"]}


# ---------------------------------------------------------------- feature_extractor.py:40-76
def fpn(feats, W, raw=None):
    def conv(x, name, k, stride=1):
        # conv2d_same (layer_utils.py:15-43): 'same' for stride 1, explicit pad + 'valid' else
        return ops.conv2d(x, W["fpn/%s/kernel" % name], stride, "SAME" if stride == 1 else "EXPLICIT")

    x = conv(feats["c5"], "lateral5", 1)
    lat = {"x5": x}
    p = {"p5": conv(x, "p5", 3)}
    p["p6"] = conv(feats["c5"], "p6", 3, 2)
    p["p7"] = conv(ops.relu(p["p6"]), "p7", 3, 2)
    for i in (4, 3):
        lateral = conv(feats["c%d" % i], "lateral%d" % i, 1)
        x = ops.upsample2_add(x, lateral)
        lat["x%d" % i] = x
        p["p%d" % i] = conv(x, "p%d" % i, 3)
    if raw is not None:
        raw.update(lat)
        raw.update({k + "_raw": v for k, v in p.items()})
    return [_bn(p["p%d" % i], W, "fpn/p%d_batch_norm" % i, "relu") for i in range(3, 8)]


# ---------------------------------------------------------------- box_predictor.py:36-155
def _tower(x, W, net, level):
    for i in range(4):
        x = ops.conv2d(x, W["%s/conv3x3_%d/kernel" % (net, i)], 1, "SAME")
        x = _bn(x, W, "%s/batch_norm_%d_for_level_%d" % (net, i, level), "relu")
    return x


def box_predictor(ps, W, num_classes, towers=None):
    A = 6
    enc, cls = [], []
    for level, p in enumerate(ps, 3):
        t = _tower(p, W, "box_net", level)
        y = ops.bias_add(ops.conv2d(t, W["box_net/encoded_boxes/kernel"], 1, "SAME"),
                         W["box_net/encoded_boxes/bias"])
        B, h, w, _ = y.shape
        enc.append(y.reshape(B, h * w * A, 4))               # reshape_and_concatenate :67-104
        t2 = _tower(p, W, "class_net", level)
        z = ops.bias_add(ops.conv2d(t2, W["class_net/logits/kernel"], 1, "SAME"),
                         W["class_net/logits/bias"])
        cls.append(z.reshape(B, h * w * A, num_classes))
        if towers is not None:
            towers["box_tower_%d" % level] = t
            towers["class_tower_%d" % level] = t2
    return np.concatenate(enc, axis=1), np.concatenate(cls, axis=1)


# ---------------------------------------------------------------- create_pb.py + model.py PREDICT
def forward(images_u8, W, params, keep=None):
    """images_u8 [B,H,W,3] uint8, any size.  create_pb.py:42-47: to_float ->
    resize_keeping_aspect_ratio(min_dimension, 128) (pipeline.py:138-194; identity when H, W
    are multiples of 128 with min(H, W) == min_dimension) -> *1/255; then model_fn PREDICT.
    Returns the graph outputs dict (model.py:70-73) plus, when `keep` is a dict, every
    intermediate needed by the stage parity tests."""
    B, H0, W0, _ = images_u8.shape
    dims, box_scaler = ops.resize_dims(H0, W0, params["min_dimension"], 128)
    xf = ops.resize_pad(images_u8.astype(np.float32), dims)
    H, Wd = xf.shape[1], xf.shape[2]
    assert H % 128 == 0 and Wd % 128 == 0
    x = ops.preprocess_f(xf)
    inter = {} if keep is None else keep
    if params["backbone"] == "mobilenet":
        feats = mobilenet_v1(x, W, inter)
    else:
        feats = shufflenet_v2(x, W, inter)
    inter.update(feats)
    ps = fpn(feats, W, inter)
    for i, p in enumerate(ps, 3):
        inter["p%d" % i] = p
    codes, logits = box_predictor(ps, W, params["num_classes"], inter)
    inter["encoded_boxes"], inter["class_predictions"] = codes, logits
    anc = ops.anchors(H, Wd)
    boxes, labels, scores, num = ops.postprocess(
        logits, codes, anc, params["score_threshold"], params["iou_threshold"],
        params["max_boxes_per_class"], box_scaler)
    return {"boxes": boxes, "labels": labels, "scores": scores, "num_boxes": num}


def detector_call(outputs, score_threshold=0.1):
    """inference/detector.py:54-60 for image 0 of `outputs`."""
    n = outputs["num_boxes"][0]
    keep = outputs["scores"][0][:n] > score_threshold
    return (outputs["boxes"][0][:n][keep], outputs["labels"][0][:n][keep],
            outputs["scores"][0][:n][keep])
