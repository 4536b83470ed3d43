"""ctypes binding of oracle/libssd_oracle.so (CPU ORACLE -- test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product path (single-shot-detector_amd/) never does.  PARITY UNPINNED: see
the header of ssd_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSD_ORACLE_LIB: another build of the same source (the AddressSanitizer + UBSan build of `make san`, scripts/oracle_sanitize.sh)
_LIB_PATH = os.environ.get("SSD_ORACLE_LIB") or os.path.join(_HERE, "libssd_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.c_int
_i64 = ctypes.c_int64


def build(force=False):
    """Compile the C restatement with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "ssd_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libssd_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_sigmoid.restype = ctypes.c_float
        _lib.orc_sigmoid.argtypes = [ctypes.c_float]
        _lib.orc_num_anchors.restype = _i
        _lib.orc_anchors_ex.restype = ctypes.c_longlong
    return _lib


def set_num_threads(n):
    lib().orc_set_num_threads(_i(int(n)))


def get_max_threads():
    return int(lib().orc_get_max_threads())


# Parity risk register (ssd_oracle.c: five single-sourced TF semantics, each with an oracle-side alternate reading; 0 = default)
ALTERNATES = {"nms_tie": 0, "fast_exp": 1, "round": 2, "resize": 3, "bn_form": 4}


def set_alternate(name, value):
    if lib().orc_set_alternate(_i(ALTERNATES[name]), _i(int(value))) != 0:
        raise ValueError(name)


def get_alternate(name):
    return int(lib().orc_get_alternate(_i(ALTERNATES[name])))


def _p(a, ty=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


def preprocess(img_u8):
    img_u8 = _c(img_u8, np.uint8)
    out = np.empty(img_u8.shape, np.float32)
    lib().orc_preprocess(_p(img_u8, ctypes.c_uint8), _i64(img_u8.size), _p(out))
    return out


def resize_dims(height, width, min_dimension=640, divisor=128):
    """pipeline.py:138-194 size arithmetic -> (new_h, new_w, pad_h, pad_w), box_scaler[4]."""
    dims = (ctypes.c_int * 4)()
    bs = np.empty(4, np.float32)
    lib().orc_resize_dims(_i(height), _i(width), _i(min_dimension), _i(divisor), dims, _p(bs))
    return tuple(dims), bs


def resize_pad(img_f, dims):
    """NN resize of a float image batch [B,H,W,C] to (new_h,new_w) + zero pad bottom/right."""
    img_f = _c(img_f)
    B, H, W, C = img_f.shape
    nh, nw, ph, pw = dims
    out = np.empty((B, nh + ph, nw + pw, C), np.float32)
    lib().orc_resize_pad(_p(img_f), _i(B), _i(H), _i(W), _i(C), _i(nh), _i(nw), _i(ph), _i(pw), _p(out))
    return out


def preprocess_f(img_f):
    img_f = _c(img_f)
    out = np.empty_like(img_f)
    lib().orc_preprocess_f(_p(img_f), _i64(img_f.size), _p(out))
    return out


def out_size(n, k, stride, mode):
    """mode 'SAME' (TF) or 'EXPLICIT' (layer_utils.py:26-43: pad (k-1)//2 both sides, VALID)."""
    if mode == "SAME":
        o = -(-n // stride)
        pad_total = max((o - 1) * stride + k - n, 0)
        return o, pad_total // 2
    pad_beg = (k - 1) // 2
    pad_end = (k - 1) - pad_beg
    return (n + pad_beg + pad_end - k) // stride + 1, pad_beg


def conv2d(x, w, stride=1, mode="SAME", scalar=False):
    """x [B,H,W,Cin], w [k,k,Cin,Cout] (HWIO)."""
    x, w = _c(x), _c(w)
    B, H, W, Cin = x.shape
    k, _, cin2, Cout = w.shape
    assert cin2 == Cin
    OH, pb = out_size(H, k, stride, mode)
    OW, pb2 = out_size(W, k, stride, mode)
    assert pb == pb2
    out = np.empty((B, OH, OW, Cout), np.float32)
    fn = lib().orc_conv2d_scalar if scalar else lib().orc_conv2d
    fn(_p(x), _i(B), _i(H), _i(W), _i(Cin), _p(w), _i(k), _i(Cout), _i(stride), _i(pb),
       _i(OH), _i(OW), _p(out))
    return out


def depthwise3x3(x, w, stride=1):
    """x [B,H,W,C], w [3,3,C,1] (tf.nn.depthwise_conv2d, padding SAME)."""
    x, w = _c(x), _c(w)
    B, H, W, C = x.shape
    assert w.shape == (3, 3, C, 1)
    OH, pb = out_size(H, 3, stride, "SAME")
    OW, _ = out_size(W, 3, stride, "SAME")
    out = np.empty((B, OH, OW, C), np.float32)
    lib().orc_depthwise3x3(_p(x), _i(B), _i(H), _i(W), _i(C), _p(w), _i(stride), _i(pb),
                           _i(OH), _i(OW), _p(out))
    return out


def bn_scale(gamma, var, eps=1e-3):
    gamma, var = _c(gamma), _c(var)
    sf = np.empty_like(gamma)
    lib().orc_bn_scale(_p(gamma), _p(var), _i(gamma.size), ctypes.c_float(eps), _p(sf))
    return sf


ACT = {None: 0, "none": 0, "relu": 1, "relu6": 2}


def bn_act(x, gamma, beta, mean, var, act, eps=1e-3):
    """In-place on a fresh copy: y = act((x - mean) * (gamma * rsqrt(var+eps)) + beta)."""
    x = _c(x).copy()
    C = x.shape[-1]
    sf = bn_scale(gamma, var, eps)
    lib().orc_bn_act(_p(x), _i64(x.size // C), _i(C), _p(_c(mean)), _p(sf), _p(_c(beta)),
                     _i(ACT[act]))
    return x


def bias_add(x, bias):
    x = _c(x).copy()
    C = x.shape[-1]
    lib().orc_bias_add(_p(x), _i64(x.size // C), _i(C), _p(_c(bias)))
    return x


def relu(x):
    x = _c(x)
    out = np.empty_like(x)
    lib().orc_relu(_p(x), _i64(x.size), _p(out))
    return out


def maxpool3x3s2(x):
    x = _c(x)
    B, H, W, C = x.shape
    out = np.empty((B, (H + 1) // 2, (W + 1) // 2, C), np.float32)
    lib().orc_maxpool3x3s2(_p(x), _i(B), _i(H), _i(W), _i(C), _p(out))
    return out


def concat_shuffle_split(x, y):
    x, y = _c(x), _c(y)
    D = x.shape[-1]
    xo, yo = np.empty_like(x), np.empty_like(y)
    lib().orc_concat_shuffle_split(_p(x), _p(y), _i64(x.size // D), _i(D), _p(xo), _p(yo))
    return xo, yo


def upsample2_add(coarse, lateral):
    coarse, lateral = _c(coarse), _c(lateral)
    B, h, w, C = coarse.shape
    assert lateral.shape == (B, 2 * h, 2 * w, C)
    out = np.empty_like(lateral)
    lib().orc_upsample2_add(_p(coarse), _p(lateral), _i(B), _i(h), _i(w), _i(C), _p(out))
    return out


def anchors(H, W):
    n = lib().orc_num_anchors(_i(H), _i(W))
    out = np.empty((n, 4), np.float32)
    lib().orc_anchors(_i(H), _i(W), _p(out))
    return out


def anchors_ex(H, W, strides, scales, scale_multipliers, aspect_ratios):
    """AnchorGenerator(strides, scales, scale_multipliers, aspect_ratios)(H, W) (anchor_generator.py:13-120)."""
    n = len(strides)
    st = (ctypes.c_int * n)(*[int(s) for s in strides])
    sc = (ctypes.c_double * n)(*[float(s) for s in scales])
    mu = (ctypes.c_double * len(scale_multipliers))(*[float(m) for m in scale_multipliers])
    ar = (ctypes.c_double * len(aspect_ratios))(*[float(a) for a in aspect_ratios])
    args = (_i(H), _i(W), _i(n), st, sc, _i(len(mu)), mu, _i(len(ar)), ar)
    cnt = lib().orc_anchors_ex(*args, None)
    out = np.empty((cnt, 4), np.float32)
    lib().orc_anchors_ex(*args, _p(out))
    return out


def sigmoid(x):
    x = _c(x)
    flat = x.ravel()
    out = np.empty_like(flat)
    f = lib().orc_sigmoid
    for j in range(flat.size):
        out[j] = f(float(flat[j]))
    return out.reshape(x.shape)


def decode_clip(codes, anc):
    codes, anc = _c(codes), _c(anc)
    out = np.empty_like(codes)
    lib().orc_decode_clip(_p(codes), _p(anc), _i64(codes.shape[0]), _p(out))
    return out


def iou_greater(bi, bj, thr):
    return bool(lib().orc_iou_greater(_p(_c(bi)), _p(_c(bj)), ctypes.c_float(thr)))


def nms(boxes, scores, max_out, iou_thr, score_thr):
    """tf.image.non_max_suppression (TF r1.12 NonMaxSuppressionV3) -> selected indices."""
    boxes, scores = _c(boxes), _c(scores)
    n = boxes.shape[0]
    sel = np.zeros(max(max_out, 1), np.int32)
    k = lib().orc_nms(_p(boxes), _p(scores), _i64(1), _i(n), _i(max_out), ctypes.c_float(iou_thr),
                      ctypes.c_float(score_thr), _p(sel, ctypes.c_int))
    return sel[:k].copy()


def postprocess(logits, codes, anc, score_thr, iou_thr, max_per_class, box_scaler=None):
    """logits [B,N,C], codes [B,N,4], anchors [N,4] -> padded graph outputs
    (boxes [B,C*m,4], labels [B,C*m] i32, scores [B,C*m], num_boxes [B] i32)."""
    logits, codes, anc = _c(logits), _c(codes), _c(anc)
    B, N, C = logits.shape
    total = C * max_per_class
    bs = _c(np.ones(4, np.float32) if box_scaler is None else box_scaler)
    boxes = np.empty((B, total, 4), np.float32)
    scores = np.empty((B, total), np.float32)
    labels = np.empty((B, total), np.int32)
    num = np.empty((B,), np.int32)
    lib().orc_postprocess(_p(logits), _p(codes), _p(anc), _i(B), _i(N), _i(C),
                          ctypes.c_float(score_thr), ctypes.c_float(iou_thr), _i(max_per_class),
                          _p(bs), _p(boxes), _p(scores), _p(labels, ctypes.c_int32),
                          _p(num, ctypes.c_int32))
    return boxes, labels, scores, num
