"""Weight container: {TF variable name -> float32 ndarray}, the names and shapes of the
reference's frozen graph (SURVEY.md 8a "Weights"; mobilenet_v1.py:22-66,
shufflenet_v2.py:25-136, feature_extractor.py:55-74, box_predictor.py:47-154).

The on-disk format is a numpy .npz keyed by those names.  (A TF `.pb` importer is a later
row of the scope table; the weights behind README.md:12 are not reachable offline.)
"""
import math

import numpy as np

MOBILENET_LAYERS = [(1, 64), (2, 128), (1, 128), (2, 256), (1, 256), (2, 512), (1, 512),
                    (1, 512), (1, 512), (1, 512), (1, 512), (2, 1024), (1, 1024)]
SHUFFLENET_DEPTHS = {0.5: 48, 1.0: 116, 1.5: 176, 2.0: 224}
_BN = ("gamma", "beta", "moving_mean", "moving_variance")


def _bn(shapes, scope, c):
    for n in _BN:
        shapes["%s/%s" % (scope, n)] = (c,)


def variable_shapes(params):
    """Ordered dict name -> shape for the configured architecture."""
    shapes = {}
    nc = params["num_classes"]
    if params["backbone"] == "mobilenet":
        dm = params["depth_multiplier"]

        def depth(x):
            return max(int(x * dm), 8)

        c = depth(32)
        shapes["MobilenetV1/Conv2d_0/weights"] = (3, 3, 3, c)
        _bn(shapes, "MobilenetV1/Conv2d_0/BatchNorm", c)
        feats = {}
        for i, (_s, f) in enumerate(MOBILENET_LAYERS, 1):
            s = "MobilenetV1/Conv2d_%d_depthwise" % i
            shapes[s + "/depthwise_weights"] = (3, 3, c, 1)
            _bn(shapes, s + "/BatchNorm", c)
            s = "MobilenetV1/Conv2d_%d_pointwise" % i
            shapes[s + "/weights"] = (1, 1, c, depth(f))
            c = depth(f)
            _bn(shapes, s + "/BatchNorm", c)
            feats[i] = c
        cs = (feats[5], feats[11], feats[13])
    else:
        d0 = SHUFFLENET_DEPTHS[float(params["depth_multiplier"])]
        shapes["ShuffleNetV2/Conv1/weights"] = (3, 3, 3, 24)
        _bn(shapes, "ShuffleNetV2/Conv1/batch_norm", 24)
        cin, out, cs = 24, d0, []
        for st, units in zip((2, 3, 4), (4, 8, 4)):
            d = out // 2
            u = "ShuffleNetV2/Stage%d/unit_1" % st
            shapes[u + "/conv1x1_before/weights"] = (1, 1, cin, cin)
            _bn(shapes, u + "/conv1x1_before/batch_norm", cin)
            shapes[u + "/depthwise/depthwise_weights"] = (3, 3, cin, 1)
            _bn(shapes, u + "/depthwise/batch_norm", cin)
            shapes[u + "/conv1x1_after/weights"] = (1, 1, cin, d)
            _bn(shapes, u + "/conv1x1_after/batch_norm", d)
            shapes[u + "/second_branch/depthwise/depthwise_weights"] = (3, 3, cin, 1)
            _bn(shapes, u + "/second_branch/depthwise/batch_norm", cin)
            shapes[u + "/second_branch/conv1x1_after/weights"] = (1, 1, cin, d)
            _bn(shapes, u + "/second_branch/conv1x1_after/batch_norm", d)
            for j in range(2, units + 1):
                u = "ShuffleNetV2/Stage%d/unit_%d" % (st, j)
                shapes[u + "/conv1x1_before/weights"] = (1, 1, d, d)
                _bn(shapes, u + "/conv1x1_before/batch_norm", d)
                shapes[u + "/depthwise/depthwise_weights"] = (3, 3, d, 1)
                _bn(shapes, u + "/depthwise/batch_norm", d)
                shapes[u + "/conv1x1_after/weights"] = (1, 1, d, d)
                _bn(shapes, u + "/conv1x1_after/batch_norm", d)
            cs.append(out)
            cin, out = out, out * 2
        fin = 2048 if float(params["depth_multiplier"]) == 2.0 else 1024
        shapes["ShuffleNetV2/Conv5/weights"] = (1, 1, cin, fin)
        _bn(shapes, "ShuffleNetV2/Conv5/batch_norm", fin)
        cs = (cs[0], cs[1], fin)
    for i, c in zip((3, 4, 5), cs):
        shapes["fpn/lateral%d/kernel" % i] = (1, 1, c, 256)
    for i in range(3, 8):
        shapes["fpn/p%d/kernel" % i] = (3, 3, cs[2] if i == 6 else 256, 256)
        _bn(shapes, "fpn/p%d_batch_norm" % i, 256)
    for net, cout in (("box_net", 24), ("class_net", 6 * nc)):
        for i in range(4):
            shapes["%s/conv3x3_%d/kernel" % (net, i)] = (3, 3, 256, 256)
            for l in range(3, 8):
                _bn(shapes, "%s/batch_norm_%d_for_level_%d" % (net, i, l), 256)
        last = "encoded_boxes" if net == "box_net" else "logits"
        shapes["%s/%s/kernel" % (net, last)] = (3, 3, 256, cout)
        shapes["%s/%s/bias" % (net, last)] = (cout,)
    return shapes


def synthetic_weights(params, seed=0, logits_bias=None, head_std=0.01):
    """Deterministic random-init weights of the configured architecture (there are no
    pretrained weights offline).  Convs ~ N(0, 2/fan_in), BN gamma U[0.5,1.5], beta
    N(0,0.1), mean N(0,0.1), var U[0.5,1.5] keep activations O(1) through the net
    (SURVEY.md 8d config 2).  The two final head convs follow the reference's
    initialisers (box_predictor.py:121-130,148-154): N(0, 0.01) kernels, zero box bias,
    logits bias -log(99) unless `logits_bias` is given (a higher bias makes the random
    net emit detections, which the NMS parity tests need).
    ShuffleNet has no clipping activation and its depthwise layers no activation at all
    (shufflenet_v2.py:121,131,135), so two corrections keep ITS activations O(1) too (without them
    c5 reaches 1e4 and the logits a standard deviation of 1e3): the 1x1 after a depthwise layer sees
    an un-rectified input and is drawn with variance 1/fan_in, and every batch-norm gamma is scaled
    by 0.9167 = 1/sqrt(E[gamma^2] E[1/var]) (unit gain on average).  Both are applied after the draw:
    the random stream, and with it every MobileNet weight, is unchanged."""
    rng = np.random.default_rng(seed)
    W = {}
    for name, shape in variable_shapes(params).items():
        leaf = name.rsplit("/", 1)[1]
        if leaf == "gamma" or leaf == "moving_variance":
            v = rng.uniform(0.5, 1.5, shape)
        elif leaf == "beta" or leaf == "moving_mean":
            v = rng.normal(0.0, 0.1, shape)
        elif leaf == "bias":
            if name.startswith("class_net"):
                v = np.full(shape, -math.log(99.0) if logits_bias is None else logits_bias)
            else:
                v = np.zeros(shape)
        elif name.endswith("encoded_boxes/kernel") or name.endswith("logits/kernel"):
            v = rng.normal(0.0, head_std, shape)
        elif leaf == "depthwise_weights":
            v = rng.normal(0.0, math.sqrt(2.0 / 9.0), shape)
        else:
            fan_in = shape[0] * shape[1] * shape[2]
            v = rng.normal(0.0, math.sqrt(2.0 / fan_in), shape)
        if name.startswith("ShuffleNetV2/"):
            if name.endswith("conv1x1_after/weights"):
                v = v / math.sqrt(2.0)
            elif leaf == "gamma":
                v = v * 0.9167
        W[name] = np.ascontiguousarray(v, dtype=np.float32)
    return W


def save_weights(path, W):
    np.savez(path, **{k.replace("/", "|"): v for k, v in W.items()})


def load_weights(path):
    with np.load(path) as z:
        return {k.replace("|", "/"): np.ascontiguousarray(z[k], dtype=np.float32) for k in z.files}
