"""A SECOND, independent CPU restatement of the reference's inference graph, written against torch's own operators
(F.conv2d, F.max_pool2d, F.interpolate, tensor arithmetic) instead of oracle/ssd_oracle.c.  It shares no code and no
layout with the C oracle: NCHW tensors like the reference's shipped DATA_FORMAT (constants.py:9), torch's padding
machinery for the windows, torch's nearest-neighbour interpolation for resize / upsample, a sort-based greedy NMS.
Test infrastructure only (tests/test_oracle_second_opinion.py): it checks the oracle's padding, ordering, layout and
NMS decisions -- NOT bit patterns (torch's convolutions sum in their own order) -- and is itself unpinned against
TensorFlow: the reference cannot run here.  Citations are into the reference repository."""
import math

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-3          # constants.py BATCH_NORM_EPSILON


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _w(k):                                  # HWIO -> OIHW
    return _t(k).permute(3, 2, 0, 1).contiguous()


def same_pad(x, k, stride):
    """tf padding='SAME': out = ceil(n / s), pad_total = max((out - 1) s + k - n, 0), the smaller half in front."""
    n_h, n_w = x.shape[2], x.shape[3]
    pads = []
    for n in (n_w, n_h):                    # F.pad wants the last dimension first
        out = -(-n // stride)
        tot = max((out - 1) * stride + k - n, 0)
        pads += [tot // 2, tot - tot // 2]
    return pads


def conv_same(x, kernel, stride=1):
    """slim.conv2d / tf.layers.conv2d(padding='same') (mobilenet_v1.py:49,66; shufflenet_v2.py:50; box_predictor.py:117-130)."""
    k = kernel.shape[0]
    return F.conv2d(F.pad(x, same_pad(x, k, stride)), _w(kernel), stride=stride)


def conv2d_same_fixed(x, kernel, stride):
    """layer_utils.conv2d_same (:15-43): stride 1 -> 'same'; stride > 1 -> explicit pad (k-1)//2 both sides, then 'valid'."""
    k = kernel.shape[0]
    if stride == 1:
        return conv_same(x, kernel, 1)
    pb = (k - 1) // 2
    pe = (k - 1) - pb
    return F.conv2d(F.pad(x, [pb, pe, pb, pe]), _w(kernel), stride=stride)


def depthwise(x, kernel, stride):
    """tf.nn.depthwise_conv2d, weights [k,k,C,1], 'SAME' (depthwise_conv.py:23)."""
    C = kernel.shape[2]
    w = _t(kernel).permute(2, 3, 0, 1).contiguous()             # [C,1,k,k]
    return F.conv2d(F.pad(x, same_pad(x, 3, stride)), w, stride=stride, groups=C)


def bn(x, W, scope, act):
    """tf.layers.batch_normalization / slim batch_norm, inference form, then the activation."""
    g, b = _t(W[scope + "/gamma"]), _t(W[scope + "/beta"])
    m, v = _t(W[scope + "/moving_mean"]), _t(W[scope + "/moving_variance"])
    sh = (1, -1, 1, 1)
    y = (x - m.view(sh)) * (g / torch.sqrt(v + EPS)).view(sh) + b.view(sh)
    if act == "relu":
        y = torch.relu(y)
    elif act == "relu6":
        y = torch.clamp(y, 0.0, 6.0)
    return y


def maxpool_same(x):
    """slim.max_pool2d(3, stride 2, 'SAME') (shufflenet_v2.py:51-54): padded cells never win."""
    return F.max_pool2d(F.pad(x, same_pad(x, 3, 2), value=float("-inf")), 3, 2)


def mobilenet(x, W, feats):
    layers = [(1, 64), (2, 128), (1, 128), (2, 256), (1, 256), (2, 512), (1, 512), (1, 512), (1, 512), (1, 512), (1, 512),
              (2, 1024), (1, 1024)]                               # mobilenet_v1.py:52-58
    s = "MobilenetV1/Conv2d_0"
    x = bn(conv_same(x, W[s + "/weights"], 2), W, s + "/BatchNorm", "relu6")
    for i, (stride, _f) in enumerate(layers, 1):
        s = "MobilenetV1/Conv2d_%d_depthwise" % i
        x = bn(depthwise(x, W[s + "/depthwise_weights"], stride), W, s + "/BatchNorm", "relu6")
        s = "MobilenetV1/Conv2d_%d_pointwise" % i
        x = bn(conv_same(x, W[s + "/weights"], 1), W, s + "/BatchNorm", "relu6")
        feats["Conv2d_%d_pointwise" % i] = x
    return {"c3": feats["Conv2d_5_pointwise"], "c4": feats["Conv2d_11_pointwise"], "c5": feats["Conv2d_13_pointwise"]}


def shuffle_split(x, y):
    """concat_shuffle_split (shufflenet_v2.py:94-115) as the reference writes it: stack -> transpose -> reshape -> split."""
    B, C, H, W = x.shape
    z = torch.stack([x, y], dim=1)                                # [B,2,C,H,W]
    z = z.transpose(1, 2).reshape(B, 2 * C, H, W)                 # channel 2c+g
    return z[:, :C], z[:, C:]


def shufflenet(x, W, feats):
    def pw(x, scope):
        return bn(conv_same(x, W[scope + "/weights"], 1), W, scope + "/batch_norm", "relu")

    def dw(x, scope, stride):
        return bn(depthwise(x, W[scope + "/depthwise_weights"], stride), W, scope + "/batch_norm", None)

    s = "ShuffleNetV2/Conv1"
    x = bn(conv_same(x, W[s + "/weights"], 2), W, s + "/batch_norm", "relu")
    x = maxpool_same(x)
    for stage, units in ((2, 4), (3, 8), (4, 4)):
        u = "ShuffleNetV2/Stage%d/unit_1" % stage
        y = pw(dw(pw(x, u + "/conv1x1_before"), u + "/depthwise", 2), u + "/conv1x1_after")
        x = pw(dw(x, u + "/second_branch/depthwise", 2), u + "/second_branch/conv1x1_after")
        for j in range(2, units + 1):
            x, y = shuffle_split(x, y)
            u = "ShuffleNetV2/Stage%d/unit_%d" % (stage, j)
            x = pw(dw(pw(x, u + "/conv1x1_before"), u + "/depthwise", 1), u + "/conv1x1_after")
        x = torch.cat([x, y], dim=1)
        feats["Stage%d" % stage] = x
    x = pw(x, "ShuffleNetV2/Conv5")
    return {"c3": feats["Stage2"], "c4": feats["Stage3"], "c5": x}


def fpn(f, W):
    def conv(x, name, stride=1):
        return conv2d_same_fixed(x, W["fpn/%s/kernel" % name], stride)
    x = conv(f["c5"], "lateral5")
    p = {5: conv(x, "p5"), 6: conv(f["c5"], "p6", 2)}
    p[7] = conv(torch.relu(p[6]), "p7", 2)
    for i in (4, 3):
        x = F.interpolate(x, scale_factor=2, mode="nearest") + conv(f["c%d" % i], "lateral%d" % i)
        p[i] = conv(x, "p%d" % i)
    return [bn(p[i], W, "fpn/p%d_batch_norm" % i, "relu") for i in range(3, 8)]


def heads(ps, W, num_classes):
    enc, cls = [], []
    for level, p in enumerate(ps, 3):
        out = []
        for net, last in (("box_net", "encoded_boxes"), ("class_net", "logits")):
            t = p
            for i in range(4):
                t = bn(conv_same(t, W["%s/conv3x3_%d/kernel" % (net, i)], 1), W, "%s/batch_norm_%d_for_level_%d" % (net, i, level), "relu")
            y = conv_same(t, W["%s/%s/kernel" % (net, last)], 1) + _t(W["%s/%s/bias" % (net, last)]).view(1, -1, 1, 1)
            out.append(y)
        B, _c, h, w = out[0].shape
        # reshape_and_concatenate (box_predictor.py:83-99): NCHW -> NHWC -> [B, h*w*A, .]
        enc.append(out[0].permute(0, 2, 3, 1).reshape(B, h * w * 6, 4))
        cls.append(out[1].permute(0, 2, 3, 1).reshape(B, h * w * 6, num_classes))
    return torch.cat(enc, 1), torch.cat(cls, 1)


def resize_keeping_aspect_ratio(img_f, min_dimension, divisor=128):
    """pipeline.py:138-194 on a float NCHW batch: NN resize (min side -> min_dimension), zero pad bottom / right."""
    H, W = img_f.shape[2], img_f.shape[3]
    scale = np.float32(min_dimension / min(H, W))
    if H >= W:
        nh = int(np.rint(np.float32(H) * scale))            # tf.round: half to even
        nw, ph, pw = min_dimension, int(math.ceil(nh / divisor)) * divisor - nh, 0
    else:
        nw = int(np.rint(np.float32(W) * scale))
        nh, pw, ph = min_dimension, int(math.ceil(nw / divisor)) * divisor - nw, 0
    r = F.interpolate(img_f, size=(nh, nw), mode="nearest")   # TF r1.12 ResizeNearestNeighbor (align_corners False): src = floor(dst * in / out)
    r = F.pad(r, [0, pw, 0, ph])
    scaler = np.array([nh / (nh + ph), nw / (nw + pw)] * 2, np.float32)
    return r, scaler


def anchors(H, W):
    """anchor_generator.py:40-170 with model.py:37-42's hyper-parameters, float32 numpy (vectorised, unlike the oracle's loops)."""
    out = []
    for stride, base in zip((8, 16, 32, 64, 128), (32, 64, 128, 256, 512)):
        h, w = int(math.ceil(H / stride)), int(math.ceil(W / stride))
        pairs = [(m, r) for m in (1.0, 1.4142) for r in (1.0, 2.0, 0.5)]
        scales = np.array([m * base for m, _ in pairs], np.float32)
        rs = np.sqrt(np.array([r for _, r in pairs], np.float32))
        hh, ww = scales / rs, scales * rs
        oy = np.float32(0.5) * (np.float32(H) - (np.float32(h) - 1) * np.float32(stride))
        ox = np.float32(0.5) * (np.float32(W) - (np.float32(w) - 1) * np.float32(stride))
        cy = np.arange(h, dtype=np.float32) * np.float32(stride) + oy
        cx = np.arange(w, dtype=np.float32) * np.float32(stride) + ox
        cyg, cxg = np.meshgrid(cy, cx, indexing="ij")
        c = np.stack([cyg, cxg], -1)[:, :, None, :]                         # [h,w,1,2]
        half = np.float32(0.5) * np.stack([hh, ww], -1)[None, None]         # [1,1,6,2]
        out.append(np.concatenate([c - half, c + half], -1).reshape(-1, 4))
    return (np.concatenate(out, 0) / np.array([H, W, H, W], np.float32)).astype(np.float32)


def decode(codes, anc):
    """box_utils.decode (:114-142) + clip (nms.py:77), float32 numpy."""
    ha, wa = anc[:, 2] - anc[:, 0], anc[:, 3] - anc[:, 1]
    cya, cxa = anc[:, 0] + np.float32(0.5) * ha, anc[:, 1] + np.float32(0.5) * wa
    ty, tx, th, tw = codes[:, 0] / np.float32(10), codes[:, 1] / np.float32(10), codes[:, 2] / np.float32(5), codes[:, 3] / np.float32(5)
    h, w = np.exp(th.astype(np.float64)).astype(np.float32) * ha, np.exp(tw.astype(np.float64)).astype(np.float32) * wa
    cy, cx = ty * ha + cya, tx * wa + cxa
    b = np.stack([cy - np.float32(0.5) * h, cx - np.float32(0.5) * w, cy + np.float32(0.5) * h, cx + np.float32(0.5) * w], 1)
    return np.clip(b, 0.0, 1.0).astype(np.float32)


def nms_sorted(boxes, scores, max_out, iou_thr, score_thr):
    """tf.image.non_max_suppression semantics, sort-based: candidates score > thr in (score desc, index asc)
    order; keep iff IoU <= thr with every kept box; IoU = 0 when either area <= 0."""
    idx = np.nonzero(scores > score_thr)[0]
    idx = idx[np.lexsort((idx, -scores[idx].astype(np.float64)))]
    kept = []
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    for i in idx:
        ok = True
        for j in kept:
            if area[i] <= 0 or area[j] <= 0:
                continue
            ih = max(np.float32(0), min(boxes[i, 2], boxes[j, 2]) - max(boxes[i, 0], boxes[j, 0]))
            iw = max(np.float32(0), min(boxes[i, 3], boxes[j, 3]) - max(boxes[i, 1], boxes[j, 1]))
            inter = np.float32(ih) * np.float32(iw)
            if inter / (area[i] + area[j] - inter) > iou_thr:
                ok = False
                break
        if ok:
            kept.append(i)
            if len(kept) == max_out:
                break
    return np.array(kept, np.int64)


def postprocess(logits, codes, anc, score_thr, iou_thr, max_per_class, box_scaler):
    """ssd.py:60 + nms.py:48-102 + model.py:67-68 for ONE image; returns (boxes, labels, scores, num) unpadded."""
    scores = (1.0 / (1.0 + np.exp(-logits.astype(np.float64)))).astype(np.float32)
    keep = scores.max(axis=1) >= score_thr                                   # nms.py:71 (>=)
    rows = np.nonzero(keep)[0]
    boxes = decode(codes[rows], anc[rows])
    ob, ol, os_ = [], [], []
    for c in range(scores.shape[1]):
        k = nms_sorted(boxes, scores[rows, c], max_per_class, np.float32(iou_thr), np.float32(score_thr))
        ob.append(boxes[k]); os_.append(scores[rows, c][k]); ol.append(np.full(len(k), c, np.int32))
    b = np.concatenate(ob, 0) / box_scaler
    return b.astype(np.float32), np.concatenate(ol), np.concatenate(os_), sum(len(x) for x in ol)


def forward(images_u8, W, params):
    """create_pb.py:42-47 + model.py PREDICT; returns stage tensors in NHWC numpy (to compare with oracle/graph.py's keep dict)
    plus the per-image detections."""
    with torch.no_grad():
        x = _t(images_u8.astype(np.float32)).permute(0, 3, 1, 2)
        x, scaler = resize_keeping_aspect_ratio(x, params["min_dimension"])
        x = x * np.float32(1.0 / 255.0)
        x = 2.0 * x - 1.0                                                     # mobilenet_v1.py:34 / shufflenet_v2.py:37
        feats = {}
        f = mobilenet(x, W, feats) if params["backbone"] == "mobilenet" else shufflenet(x, W, feats)
        ps = fpn(f, W)
        enc, cls = heads(ps, W, params["num_classes"])
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().numpy()
    st = {k: nhwc(v) for k, v in f.items()}
    for i, p in enumerate(ps, 3):
        st["p%d" % i] = nhwc(p)
    st["encoded_boxes"], st["class_predictions"] = enc.numpy(), cls.numpy()
    anc = anchors(x.shape[2], x.shape[3])
    dets = [postprocess(st["class_predictions"][b], st["encoded_boxes"][b], anc, params["score_threshold"], params["iou_threshold"],
                        params["max_boxes_per_class"], scaler) for b in range(x.shape[0])]
    return st, anc, dets
