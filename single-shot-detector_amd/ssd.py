"""Host-side mirror of detector/ssd.py, detector/anchor_generator.py and
detector/utils/nms.py on top of libssd_hip.so.  Tensors are torch CUDA (= HIP) tensors;
torch only provides the memory and the stream.
"""
import ctypes
import threading

import numpy as np

from . import _lib
from ._lib import SsdConfig, check, lib

ACT = {None: 0, "none": 0, "relu": 1, "relu6": 2}


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("single-shot-detector_amd needs an AMD GPU (HIP device): there is "
                           "no CPU path")
    return torch


def _stream(torch):
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _fp(a):
    """host float32 array (or None) -> ctypes float pointer (keeps `a` alive via return)."""
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _check_dev(torch, t, dtype, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise TypeError("%s must be a contiguous CUDA tensor of dtype %s" % (name, dtype))


# ----------------------------------------------------------------------------- stage ops
def out_size(n, k, stride, mode):
    """'SAME' (TF) or 'EXPLICIT' (conv2d_same stride>1, layer_utils.py:26-43)."""
    if mode == "SAME":
        o = -(-n // stride)
        return o, max((o - 1) * stride + k - n, 0) // 2
    pad_beg = (k - 1) // 2
    return (n + (k - 1) - k) // stride + 1, pad_beg


def conv2d(x, w, stride=1, mode="SAME", bn=None, bias=None, up=None, act=None, precision="f32"):
    """Dense 1x1 / 3x3 convolution on the MFMA kernel (ssd_conv2d / ssd_conv2d_f16x3).
    x [B,H,W,Cin] cuda f32; w HWIO numpy; bn = (mean, sf, beta) numpy; up = coarser map."""
    torch = _torch()
    _check_dev(torch, x, torch.float32, "x")
    w = np.ascontiguousarray(w, dtype=np.float32)
    B, H, W, Cin = x.shape
    k, k2, cin2, Cout = w.shape
    if k != k2 or cin2 != Cin:
        raise ValueError("kernel shape %s does not match input channels %d" % (w.shape, Cin))
    OH, pb = out_size(H, k, stride, mode)
    OW, _ = out_size(W, k, stride, mode)
    out = torch.empty((B, OH, OW, Cout), dtype=torch.float32, device=x.device)
    keep = [_fp(v) for v in (bn if bn is not None else (None, None, None))]
    kb = _fp(bias)
    if up is not None:
        _check_dev(torch, up, torch.float32, "up")
        if tuple(up.shape) != (B, OH // 2, OW // 2, Cout):
            raise ValueError("up must have shape [B, OH/2, OW/2, Cout]")
    fn = {"f32": lib().ssd_conv2d, "f16x3": lib().ssd_conv2d_f16x3}[precision]
    check(fn(_ptr(x), B, H, W, Cin, w.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
             k, Cout, stride, pb, OH, OW, keep[0][1], keep[1][1], keep[2][1], kb[1],
             _ptr(up) if up is not None else None, ACT[act], _ptr(out), _stream(torch)))
    return out


def depthwise3x3(x, w, stride=1, bn=None, act=None):
    torch = _torch()
    _check_dev(torch, x, torch.float32, "x")
    w = np.ascontiguousarray(w, dtype=np.float32)
    B, H, W, C = x.shape
    if w.shape != (3, 3, C, 1):
        raise ValueError("depthwise weights must be [3,3,C,1]")
    OH, pb = out_size(H, 3, stride, "SAME")
    OW, _ = out_size(W, 3, stride, "SAME")
    out = torch.empty((B, OH, OW, C), dtype=torch.float32, device=x.device)
    keep = [_fp(v) for v in (bn if bn is not None else (None, None, None))]
    check(lib().ssd_depthwise3x3(_ptr(x), B, H, W, C, w.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                 stride, pb, OH, OW, keep[0][1], keep[1][1], keep[2][1], ACT[act],
                                 _ptr(out), _stream(torch)))
    return out


def dw_pw(x, dw_w, stride, dw_bn, dw_act, pw_w, pw_bn, pw_act):
    """depthwise 3x3 + BN + act -> 1x1 conv + BN + act in one kernel (mobilenet_v1.py:59-67).
    bn arguments are (mean, scale_factor, beta) as for conv2d; both are required."""
    torch = _torch()
    _check_dev(torch, x, torch.float32, "x")
    dw_w = np.ascontiguousarray(dw_w, dtype=np.float32)
    pw_w = np.ascontiguousarray(pw_w, dtype=np.float32)
    B, H, W, C = x.shape
    if dw_w.shape != (3, 3, C, 1) or pw_w.shape[:3] != (1, 1, C):
        raise ValueError("weights must be [3,3,C,1] and [1,1,C,Cout]")
    Cout = pw_w.shape[3]
    OH, _ = out_size(H, 3, stride, "SAME")
    OW, _ = out_size(W, 3, stride, "SAME")
    out = torch.empty((B, OH, OW, Cout), dtype=torch.float32, device=x.device)
    kd = [_fp(v) for v in dw_bn]
    kp = [_fp(v) for v in pw_bn]
    fp = ctypes.POINTER(ctypes.c_float)
    check(lib().ssd_dw_pw(_ptr(x), B, H, W, C, dw_w.ctypes.data_as(fp), stride, kd[0][1], kd[1][1], kd[2][1],
                          ACT[dw_act], pw_w.ctypes.data_as(fp), Cout, kp[0][1], kp[1][1], kp[2][1], ACT[pw_act],
                          _ptr(out), _stream(torch)))
    return out


def first_conv(images, w, bn=None, act=None):
    torch = _torch()
    _check_dev(torch, images, torch.uint8, "images")
    w = np.ascontiguousarray(w, dtype=np.float32)
    B, H, W, three = images.shape
    if three != 3 or w.shape[:3] != (3, 3, 3):
        raise ValueError("images must be [B,H,W,3] and weights [3,3,3,Cout]")
    Cout = w.shape[3]
    out = torch.empty((B, H // 2, W // 2, Cout), dtype=torch.float32, device=images.device)
    keep = [_fp(v) for v in (bn if bn is not None else (None, None, None))]
    check(lib().ssd_first_conv(_ptr(images), B, H, W, w.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                               Cout, keep[0][1], keep[1][1], keep[2][1], ACT[act], _ptr(out),
                               _stream(torch)))
    return out


def front_block(images, w0, bn0, act0, dw_w, dw_bn, dw_act, pw_w, pw_bn, pw_act):
    """MobileNet's first three layers as the one launch the layer plan uses (front.hip): first_conv followed by dw_pw at
    stride 1, the 32-channel tensor kept on chip (mobilenet_v1.py:34-67).  Only 3 -> 32 -> 32 -> 64 channels."""
    torch = _torch()
    _check_dev(torch, images, torch.uint8, "images")
    w0 = np.ascontiguousarray(w0, dtype=np.float32)
    dw_w = np.ascontiguousarray(dw_w, dtype=np.float32)
    pw_w = np.ascontiguousarray(pw_w, dtype=np.float32)
    B, H, W, three = images.shape
    if three != 3 or w0.shape[:3] != (3, 3, 3):
        raise ValueError("images must be [B,H,W,3] and the first weights [3,3,3,C0]")
    C0 = w0.shape[3]
    if dw_w.shape != (3, 3, C0, 1) or pw_w.shape[:3] != (1, 1, C0):
        raise ValueError("weights must be [3,3,C0,1] and [1,1,C0,Cout]")
    Cout = pw_w.shape[3]
    out = torch.empty((B, H // 2, W // 2, Cout), dtype=torch.float32, device=images.device)
    k0 = [_fp(v) for v in bn0]
    kd = [_fp(v) for v in dw_bn]
    kp = [_fp(v) for v in pw_bn]
    fp = ctypes.POINTER(ctypes.c_float)
    check(lib().ssd_front_block(_ptr(images), B, H, W, w0.ctypes.data_as(fp), C0, k0[0][1], k0[1][1], k0[2][1], ACT[act0],
                                dw_w.ctypes.data_as(fp), kd[0][1], kd[1][1], kd[2][1], ACT[dw_act],
                                pw_w.ctypes.data_as(fp), Cout, kp[0][1], kp[1][1], kp[2][1], ACT[pw_act],
                                _ptr(out), _stream(torch)))
    return out


def first_conv_maxpool(images, w, bn, act=None):
    """ShuffleNet's first convolution + 3x3 stride-2 max pool as the one launch the layer plan uses (front.hip;
    shufflenet_v2.py:50-54).  Only 24 output channels, H and W multiples of 4."""
    torch = _torch()
    _check_dev(torch, images, torch.uint8, "images")
    w = np.ascontiguousarray(w, dtype=np.float32)
    B, H, W, three = images.shape
    if three != 3 or w.shape[:3] != (3, 3, 3):
        raise ValueError("images must be [B,H,W,3] and weights [3,3,3,Cout]")
    Cout = w.shape[3]
    out = torch.empty((B, H // 4, W // 4, Cout), dtype=torch.float32, device=images.device)
    keep = [_fp(v) for v in bn]
    check(lib().ssd_first_conv_maxpool(_ptr(images), B, H, W, w.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                       Cout, keep[0][1], keep[1][1], keep[2][1], ACT[act], _ptr(out), _stream(torch)))
    return out


def maxpool3x3s2(x):
    torch = _torch()
    _check_dev(torch, x, torch.float32, "x")
    B, H, W, C = x.shape
    out = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    check(lib().ssd_maxpool3x3s2(_ptr(x), B, H, W, C, _ptr(out), _stream(torch)))
    return out


def concat_shuffle_split(x, y):
    torch = _torch()
    _check_dev(torch, x, torch.float32, "x")
    _check_dev(torch, y, torch.float32, "y")
    if x.shape != y.shape:
        raise ValueError("x and y must have the same shape")
    D = x.shape[-1]
    xo, yo = torch.empty_like(x), torch.empty_like(y)
    check(lib().ssd_concat_shuffle_split(_ptr(x), _ptr(y), x.numel() // D, D, _ptr(xo), _ptr(yo),
                                         _stream(torch)))
    return xo, yo


def shuffle_conv1x1(x, y, w, bn, act="relu"):
    """concat_shuffle_split(x, y) -> conv1x1_before on the new x half (shufflenet_v2.py:94-115,119) as one kernel.
    x, y [..., D] cuda f32; w [1,1,D,Cout] numpy; bn = (mean, sf, beta)."""
    torch = _torch()
    _check_dev(torch, x, torch.float32, "x")
    _check_dev(torch, y, torch.float32, "y")
    if x.shape != y.shape:
        raise ValueError("x and y must have the same shape")
    w = np.ascontiguousarray(w, dtype=np.float32)
    D, Cout = x.shape[-1], w.shape[3]
    if w.shape[:3] != (1, 1, D):
        raise ValueError("kernel must be [1,1,D,Cout]")
    out = torch.empty(tuple(x.shape[:-1]) + (Cout,), dtype=torch.float32, device=x.device)
    keep = [_fp(v) for v in bn]
    check(lib().ssd_shuffle_conv1x1(_ptr(x), _ptr(y), x.numel() // D, D, w.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), Cout,
                                    keep[0][1], keep[1][1], keep[2][1], ACT[act], _ptr(out), _stream(torch)))
    return out


class AnchorGenerator:
    """detector/anchor_generator.py:12-120: any strides / scales / scale multipliers / aspect ratios (ssd_anchors_ex; the
    defaults are the values model.py:37-42 fixes for the exported graph)."""

    def __init__(self, strides=[8, 16, 32, 64, 128], scales=[32, 64, 128, 256, 512],
                 scale_multipliers=[1.0, 1.4142], aspect_ratios=[1.0, 2.0, 0.5]):
        assert len(strides) == len(scales)                      # anchor_generator.py:33
        self.strides, self.scales = list(strides), list(scales)
        self.scale_multipliers, self.aspect_ratios = list(scale_multipliers), list(aspect_ratios)
        self.num_anchors_per_location = len(aspect_ratios) * len(scale_multipliers)

    def _call(self, H, W, out, cap):
        n = len(self.strides)
        st = (ctypes.c_int32 * n)(*[int(s) for s in self.strides])
        sc = (ctypes.c_double * n)(*[float(s) for s in self.scales])
        mu = (ctypes.c_double * len(self.scale_multipliers))(*[float(m) for m in self.scale_multipliers])
        ar = (ctypes.c_double * len(self.aspect_ratios))(*[float(a) for a in self.aspect_ratios])
        r = lib().ssd_anchors_ex(int(H), int(W), n, st, sc, len(mu), mu, len(ar), ar, out, cap)
        if r < 0:
            check(int(r))
        return int(r)

    def __call__(self, image_height, image_width):
        """-> float32 ndarray [num_anchors, 4], normalised ymin,xmin,ymax,xmax, not clipped."""
        n = self._call(image_height, image_width, None, 0)
        out = np.empty((n, 4), np.float32)
        self._call(image_height, image_width, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n)
        ih, iw = np.float32(image_height), np.float32(image_width)
        self.num_anchors_per_feature_map = [
            int(np.ceil(ih / np.float32(s))) * int(np.ceil(iw / np.float32(s))) * self.num_anchors_per_location
            for s in self.strides]
        return out


def batch_multiclass_non_max_suppression(encoded_boxes, anchors, logits, score_threshold,
                                         iou_threshold, max_boxes_per_class, box_scaler=None):
    """detector/utils/nms.py:48-102 on the GPU (ssd_postprocess).

    encoded_boxes [B,N,4], anchors [N,4], logits [B,N,C] CUDA tensors.  Unlike the
    reference, which is handed tf.sigmoid(class_predictions), this takes the LOGITS: the
    sigmoid of ssd.py:60 is fused into the scan kernel.
    Returns boxes [B,C*m,4], scores [B,C*m], classes [B,C*m] int32, num_detections [B] int32
    (the reference's return order, nms.py:102)."""
    torch = _torch()
    for t, n in ((encoded_boxes, "encoded_boxes"), (anchors, "anchors"), (logits, "logits")):
        _check_dev(torch, t, torch.float32, n)
    B, N, C = logits.shape
    if tuple(encoded_boxes.shape) != (B, N, 4) or tuple(anchors.shape) != (N, 4):
        raise ValueError("shape mismatch between encoded_boxes, anchors and logits")
    T = C * int(max_boxes_per_class)
    dev = logits.device
    boxes = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((B, T), dtype=torch.float32, device=dev)
    classes = torch.empty((B, T), dtype=torch.int32, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    nbytes = lib().ssd_postprocess_workspace_bytes(B, N, C, int(max_boxes_per_class))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    bs = _fp(box_scaler)
    check(lib().ssd_postprocess(_ptr(logits), _ptr(encoded_boxes), _ptr(anchors), B, N, C,
                                float(score_threshold), float(iou_threshold),
                                int(max_boxes_per_class), bs[1], _ptr(boxes), _ptr(classes),
                                _ptr(scores), _ptr(num), _ptr(ws), nbytes, _stream(torch)))
    return boxes, scores, classes, num


# ----------------------------------------------------------------------------- whole graph
class Engine:
    """Owns one ssd_handle: weights + workspace on one GPU.  `forward` is the frozen
    graph's sess.run (inference/detector.py:51-52) for a batch."""

    PRECISIONS = {"f32": 0, "f16x3": 1}     # SSD_PRECISION_* of include/ssd_hip.h

    def __init__(self, params, weights, device=0, precision=None):
        """precision: "f32" (every convolution an exact fp32 fmaf chain, bit-identical to the
        oracle), "f16x3" (FPN + heads on split-fp16 operands, fp32-chain accuracy, ~1e-6 from
        the oracle) or None = params["precision"] if present, else the library default (the
        SSD_PRECISION environment variable, else f32)."""
        torch = _torch()
        self.params = dict(params)
        self.device = int(device)
        if precision is None:
            precision = self.params.get("precision")
        if precision is not None and precision not in self.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(self.PRECISIONS))
        cfg = SsdConfig(0 if params["backbone"] == "mobilenet" else 1,
                        float(params["depth_multiplier"]), int(params["num_classes"]),
                        float(params["score_threshold"]), float(params["iou_threshold"]),
                        int(params["max_boxes_per_class"]), int(params["min_dimension"]),
                        self.device)
        self._h = ctypes.c_void_p()
        torch.cuda.set_device(self.device)
        check(lib().ssd_create(ctypes.byref(cfg), ctypes.byref(self._h)))
        try:
            for name, arr in weights.items():
                a = np.ascontiguousarray(arr, dtype=np.float32)
                shape = (ctypes.c_int64 * a.ndim)(*a.shape)
                check(lib().ssd_load_weight(self._h, name.encode(),
                                            a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                            shape, a.ndim))
            check(lib().ssd_finalize(self._h))
            if precision is not None:
                check(lib().ssd_set_precision(self._h, self.PRECISIONS[precision]))
        except Exception:
            lib().ssd_destroy(self._h)
            self._h = None
            raise
        self.T = int(params["num_classes"]) * int(params["max_boxes_per_class"])
        self.one_call_detect = True      # Detector.__call__ through ssd_detect_host (False: detect_host + numpy filter)
        self._static = {}
        self._copy_streams = None
        self.zero_copy_max_batch = 1     # detect_host: up to this many images the records are written straight into pinned host memory
        self.lock = threading.RLock()       # serialises the calls on this engine (tf.Session.run is thread-safe)

    @property
    def precision(self):
        v = lib().ssd_get_precision(self._h)
        return {n: k for k, n in self.PRECISIONS.items()}[v]

    def set_precision(self, precision):
        with self.lock:
            check(lib().ssd_set_precision(self._h, self.PRECISIONS[precision]))

    def status(self):
        """Synchronises; returns and clears the handle's status word (bit 0: an f16x3 activation
        left the fp16 range and was clamped -- re-run those forwards with precision "f32")."""
        v = ctypes.c_int32()
        with self.lock:
            check(lib().ssd_status(self._h, ctypes.byref(v)))
        return v.value

    def close(self):
        if getattr(self, "_h", None):
            lib().ssd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        """ssd_set_option on this engine's handle (include/ssd_hip.h lists the keys): a kernel / schedule selector that
        does not change a result bit in mode f32.  Synchronises and drops the cached layer plans (except "plan_cache_mb",
        which only evicts down to the new budget)."""
        with self.lock:
            _lib.set_option(key, value, self._h)

    def forward(self, images, out=None, records=None):
        """images: uint8 CUDA tensor [B,H,W,3] -> (boxes [B,T,4], labels [B,T] i32,
        scores [B,T], num_boxes [B] i32) CUDA tensors; asynchronous on the current stream.
        The four outputs are views of ONE block of B fixed records (`records`: an int32 tensor [B, 6T+1] -- device memory
        or pinned host memory -- created here when not given; ssd_forward_records): the unit the data-parallel all-gather
        moves and one copy brings to the host.  `out` = four dense tensors of the caller's instead (ssd_forward).
        Thread-safe like tf.Session.run (inference/detector.py:34,52): calls on one engine are serialised (here and by the
        handle's own mutex), and the library orders the GPU work of consecutive forwards even across streams."""
        torch = _torch()
        _check_dev(torch, images, torch.uint8, "images")
        if images.dim() != 4 or images.shape[3] != 3:
            raise ValueError("images must have shape [B,H,W,3]")
        B, H, W, _ = images.shape
        if out is not None:
            boxes, labels, scores, num = out
            for t in out:
                if not t.is_contiguous():
                    raise ValueError("out: four dense tensors (use records= for the packed block)")
            with self.lock:
                check(lib().ssd_forward(self._h, _ptr(images), B, H, W, _ptr(boxes), _ptr(labels),
                                        _ptr(scores), _ptr(num), _stream(torch)))
            return boxes, labels, scores, num
        if records is None:
            records = self.new_records(B, images.device)
        if records.dtype != torch.int32 or tuple(records.shape) != (B, self.record_words) or not records.is_contiguous():
            raise ValueError("records must be a contiguous int32 tensor [B, %d]" % self.record_words)
        if not (records.is_cuda or records.is_pinned()):
            # the GPU writes through this pointer: pageable host memory would fault inside the kernel
            raise ValueError("records must live in device memory or in pinned host memory")
        with self.lock:
            check(lib().ssd_forward_records(self._h, _ptr(images), B, H, W, _ptr(records), _stream(torch)))
        return self.record_views(records)

    @property
    def record_words(self):
        """32-bit words of one image's record: boxes [T,4] f32 | scores [T] f32 | labels [T] i32 | num_boxes i32."""
        return 6 * self.T + 1

    def new_records(self, B, dev):
        return _torch().empty((B, self.record_words), dtype=_torch().int32, device=dev)

    def record_views(self, rec):
        """(boxes [B,T,4] f32, labels [B,T] i32, scores [B,T] f32, num_boxes [B] i32) as VIEWS of a record block [B, 6T+1]
        (torch tensor or numpy array): no copy."""
        T = self.T
        B = rec.shape[0]
        if isinstance(rec, np.ndarray):
            return (rec[:, :4 * T].view(np.float32).reshape(B, T, 4), rec[:, 5 * T:6 * T], rec[:, 4 * T:5 * T].view(np.float32),
                    rec[:, 6 * T])
        torch = _torch()
        return (rec[:, :4 * T].view(torch.float32).unflatten(1, (T, 4)), rec[:, 5 * T:6 * T],
                rec[:, 4 * T:5 * T].view(torch.float32), rec[:, 6 * T])

    def _new_outputs(self, torch, B, dev):
        """(record block [B, 6T+1], its four views)."""
        block = self.new_records(B, dev)
        return block, self.record_views(block)

    def _out_slot(self, B, index=0):
        """Persistent result buffers of one BATCH SIZE: the device record block, its pinned host copy and numpy / tensor views
        of both.  Independent of the image size: a mix of sizes through one engine (inference/evaluate_on_COCO.ipynb:125-150)
        reuses one slot."""
        torch = _torch()
        slot = self._static.get(("out", B, index))
        if slot is None:
            dev = "cuda:%d" % self.device
            block, views = self._new_outputs(torch, B, dev)
            pin_out = torch.empty(block.shape, dtype=torch.int32).pin_memory()
            slot = {"block": block, "views": views, "pin_out": pin_out, "host": self.record_views(pin_out.numpy())}
            while sum(1 for k in self._static if k[0] == "out") >= 32:
                self._static.pop(next(k for k in self._static if k[0] == "out"))
            self._static[("out", B, index)] = slot
        return slot

    def _in_slot(self, shape, index=0, pinned=False):
        """Persistent image buffers, grow-only and flat (powers of two from 2 MiB): (device image, pinned staging image or
        None) viewed as `shape`.  A new image size allocates only when it is larger than every one before it."""
        torch = _torch()
        n = 1
        for d in shape:
            n *= int(d)
        slot = self._static.setdefault(("in", index), {"cap": 0, "dev": None, "pin": None, "ev": None, "events": None})
        if n > slot["cap"] or (pinned and slot["pin"] is None):
            cap = max(slot["cap"], 2 << 20)
            while cap < n:
                cap <<= 1
            # (whatever still reads the old buffers: this is the rare path)
            torch.cuda.synchronize(self.device)
            slot["dev"] = torch.empty((cap,), dtype=torch.uint8, device="cuda:%d" % self.device)
            slot["pin"] = torch.empty((cap,), dtype=torch.uint8).pin_memory() if pinned or slot["pin"] is not None else None
            slot["cap"] = cap
        dev_in = slot["dev"][:n].view(shape)
        pin_in = slot["pin"][:n].view(shape) if slot["pin"] is not None else None
        return slot, dev_in, pin_in

    def forward_cached(self, images):
        """Serving form of `forward`: `images` (uint8 numpy array or tensor [B,H,W,3]) is copied
        into a persistent device buffer and the outputs live in persistent buffers too (one set per batch size), so
        steady-state calls allocate nothing.  The returned tensors are overwritten by the next call with this batch
        size: consume (e.g. `.cpu()`) before calling again."""
        torch = _torch()
        if isinstance(images, np.ndarray):
            images = torch.from_numpy(np.ascontiguousarray(images))
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3:
            raise ValueError("images must be uint8 with shape [B,H,W,3]")
        with self.lock:
            slot, dev_in, _ = self._in_slot(tuple(images.shape))
            out = self._out_slot(int(images.shape[0]))
            cur = torch.cuda.current_stream()
            if slot["ev"] is not None:          # another thread's stream may still be using these buffers
                cur.wait_event(slot["ev"])
            else:
                slot["ev"] = torch.cuda.Event()
            dev_in.copy_(images, non_blocking=True)
            res = self.forward(dev_in, records=out["block"])
            slot["ev"].record(cur)
            return res

    def detect_host(self, images):
        """The boundary's own form (inference/detector.py:51-52: a host ndarray in, numpy out on every call), at its best
        case: host uint8 [B,H,W,3] -> pinned staging -> HBM, forward, ONE device-to-host copy of the packed outputs into
        pinned memory, one stream synchronisation.  Returns numpy VIEWS of that pinned block (boxes, labels, scores,
        num_boxes), valid until the next call with this batch size: copy or reduce them before calling again."""
        torch = _torch()
        images = np.asarray(images)
        if images.dtype != np.uint8 or images.ndim != 4 or images.shape[3] != 3:
            raise ValueError("images must be a uint8 array of shape [B, height, width, 3]")
        with self.lock:
            src = np.ascontiguousarray(images)
            B, H, W, _ = src.shape
            slot = self._out_slot(B)
            # ssd_forward_host: staging copy + upload in pieces (one C loop) and the forward, on the current stream.  One image:
            # post_pack_kernel writes its 48 KB record into the pinned block itself (16-byte rows over PCIe) -- no
            # device-to-host copy behind the forward.  (Records are 48 004 bytes: those of further images are not 16-byte
            # aligned and would cross the bus in 4-byte writes -- they go through the device block and one copy.)
            zc = B <= self.zero_copy_max_batch
            rec = slot["pin_out"] if zc else slot["block"]
            check(lib().ssd_forward_host(self._h, src.ctypes.data, B, H, W, _ptr(rec), _stream(torch)))
            if not zc:
                slot["pin_out"].copy_(slot["block"], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            return slot["host"]

    def detect_one(self, image, score_threshold):
        """inference/detector.py:33-58 for one host frame [H,W,3] as ONE library call (ssd_detect_host): staging copy + upload,
        forward, the wait, and the score filter in C.  Returns fresh arrays boxes [n,4], labels [n], scores [n].  Nothing
        here depends on the frame's size: the pinned record and the scratch arrays are the engine's, the layer plan of the
        network shape the frame resizes to is the library's (kept per shape, ssd_hip.h)."""
        torch = _torch()
        H, W, _ = image.shape
        with self.lock:
            slot = self._out_slot(1)
            src = np.ascontiguousarray(image)
            sc = slot.get("one")
            if sc is None:
                T = self.T
                sc = (np.empty((T, 4), np.float32), np.empty((T,), np.int32), np.empty((T,), np.float32), ctypes.c_int32(0))
                sc = slot["one"] = sc + tuple(a.ctypes.data for a in sc[:3]) + (ctypes.byref(sc[3]), _ptr(slot["pin_out"]))
            check(lib().ssd_detect_host(self._h, src.ctypes.data, H, W, score_threshold, sc[8], sc[4], sc[5], sc[6],
                                        sc[0].shape[0], sc[7], _stream(torch)))
            n = sc[3].value
            return sc[0][:n].copy(), sc[1][:n].copy(), sc[2][:n].copy()

    MIXED_MAX = 64       # frames per mixed-size batch (their geometry travels in the first kernel's arguments)

    def network_shape(self, height, width):
        """(height, width) the network sees for a frame of this size: what frames are grouped by for a mixed-size batch."""
        return network_input_size(height, width, self.params["min_dimension"])[:2]

    def forward_mixed(self, frames, records=None):
        """Frames of DIFFERENT sizes that resize to one network shape, as ONE batch (ssd_forward_mixed).  `frames`: a list of
        uint8 CUDA tensors [H_b, W_b, 3] (copied back to back into a persistent device buffer), or a tuple (flat uint8 CUDA
        tensor, [(H_b, W_b)...], byte offsets or None) already laid out.  Returns (boxes [B,T,4], labels, scores, num_boxes) as
        views of `records` (created when not given); image b is bit for bit what frame b gives alone.  Asynchronous on the
        current stream."""
        torch = _torch()
        with self.lock:
            if isinstance(frames, tuple):
                flat, hw, offsets = frames
                _check_dev(torch, flat, torch.uint8, "frames")
            else:
                hw = [(int(f.shape[0]), int(f.shape[1])) for f in frames]
                offsets, total = [], 0
                for h_, w_ in hw:
                    offsets.append(total)
                    total += (h_ * w_ * 3 + 15) & ~15
                _, flat, _ = self._in_slot((max(total, 16),), index=4)
                for f, o, (h_, w_) in zip(frames, offsets, hw):
                    _check_dev(torch, f, torch.uint8, "frames")
                    if f.dim() != 3 or f.shape[2] != 3:
                        raise ValueError("every frame must have shape [H, W, 3]")
                    flat[o:o + h_ * w_ * 3].view(h_, w_, 3).copy_(f, non_blocking=True)
            B = len(hw)
            if records is None:
                records = self.new_records(B, flat.device)
            if records.dtype != torch.int32 or tuple(records.shape) != (B, self.record_words) or not records.is_contiguous():
                raise ValueError("records must be a contiguous int32 tensor [B, %d]" % self.record_words)
            hw_c = (ctypes.c_int32 * (2 * B))(*[v for pair in hw for v in pair])
            off_c = (ctypes.c_int64 * B)(*offsets) if offsets is not None else None
            check(lib().ssd_forward_mixed(self._h, _ptr(flat), B, hw_c, off_c, _ptr(records), _stream(torch)))
            return self.record_views(records)

    def detect_host_mixed(self, images, wait=True, index=0):
        """A list of host uint8 arrays [H_b, W_b, 3] of different sizes that share a network shape -> numpy VIEWS of the pinned
        result block (boxes, labels, scores, num_boxes), valid until the next call with this batch size: ONE call of
        ssd_forward_mixed_host (staging + one upload per frame + the batched forward), one device-to-host copy, one wait.
        wait=False: returns the result slot right after enqueuing (slot["done"]: an event behind the copy, slot["host"]: the views
        it makes valid) -- the caller stages the next batch while this one computes (Detector.detect_many; `index` picks one of
        several result slots of this batch size)."""
        torch = _torch()
        B = len(images)
        with self.lock:
            srcs = [np.ascontiguousarray(im) for im in images]
            for im in srcs:
                if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                    raise ValueError("every image must be a uint8 array of shape [height, width, 3]")
            slot = self._out_slot(B, index)
            zc = B <= self.zero_copy_max_batch
            rec = slot["pin_out"] if zc else slot["block"]
            ptrs = (ctypes.c_void_p * B)(*[im.ctypes.data for im in srcs])
            hw_c = (ctypes.c_int32 * (2 * B))(*[int(v) for im in srcs for v in im.shape[:2]])
            check(lib().ssd_forward_mixed_host(self._h, ptrs, B, hw_c, _ptr(rec), _stream(torch)))
            if not zc:
                slot["pin_out"].copy_(slot["block"], non_blocking=True)
            if not wait:
                if slot.get("done") is None:
                    slot["done"] = torch.cuda.Event()
                slot["done"].record()
                return slot
            torch.cuda.current_stream().synchronize()
            return slot["host"]

    def detect_stream(self, batches):
        """Host-fed steady-state serving: an iterable of host uint8 arrays [B,H,W,3] (any mix of sizes) -> a generator of
        (boxes, labels, scores, num_boxes) numpy arrays, in order.  Two sets of buffers and two copy streams: the
        host-to-device copy of batch k+1 and the device-to-host copy of the packed outputs of batch k-1 run under the
        compute of batch k, so the host feed costs (almost) nothing against frames already resident in HBM.  The FIRST two
        batches of a new batch shape pay for the buffer sets (pinning 2 x 32 MB of host memory takes ~0.3 s for 32 frames of
        640 x 896): results of those batches arrive late; a serving process sees it once.  Results are
        bit-identical to detect_host on the same batches (same kernels, same order)."""
        torch = _torch()
        dev = torch.device("cuda", self.device)
        if self._copy_streams is None:
            self._copy_streams = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
        s_in, s_out = self._copy_streams
        s_c = torch.cuda.current_stream(dev)
        pending = None                       # (out slot, events) of the batch whose outputs are still on their way
        k = 0

        def collect(p):
            oslot, e = p
            e["d2h"].synchronize()
            return tuple(np.array(v) for v in oslot["host"])

        for images in batches:
            images = np.asarray(images)
            if images.dtype != np.uint8 or images.ndim != 4 or images.shape[3] != 3:
                raise ValueError("every batch must be a uint8 array of shape [B, height, width, 3]")
            key = tuple(images.shape)
            j = k & 1
            islot, dev_in, pin_in = self._in_slot(key, index=1 + j, pinned=True)
            oslot = self._out_slot(key[0], index=1 + j)
            e = islot["events"]
            if e is None:
                e = islot["events"] = {n: torch.cuda.Event() for n in ("h2d", "cmp", "d2h")}
                for n in e:
                    e[n].record(s_c)
            e["h2d"].synchronize()           # the staging buffer's previous upload has left it
            e["d2h"].synchronize()           # ... and the previous results of this set have been collected below
            np.copyto(pin_in.numpy(), images)
            with torch.cuda.stream(s_in):
                s_in.wait_event(e["cmp"])    # the forward that read this device image two batches ago is over
                dev_in.copy_(pin_in, non_blocking=True)
                e["h2d"].record(s_in)
            s_c.wait_event(e["h2d"])
            self.forward(dev_in, records=oslot["block"])
            e["cmp"].record(s_c)
            with torch.cuda.stream(s_out):
                s_out.wait_event(e["cmp"])
                oslot["pin_out"].copy_(oslot["block"], non_blocking=True)
                e["d2h"].record(s_out)
            if pending is not None:
                yield collect(pending)
            pending = (oslot, e)
            k += 1
        if pending is not None:
            yield collect(pending)

    def plan_cache_stats(self):
        """ssd_plan_cache_stats: the library keeps one layer plan per network shape this engine has served."""
        v = (ctypes.c_int64 * 8)()
        with self.lock:
            check(lib().ssd_plan_cache_stats(self._h, v))
        return {"plans": v[0], "arena_bytes": v[1], "budget_bytes": v[2], "hits": v[3], "misses": v[4], "evictions": v[5],
                "last_network_shape": [v[6], v[7]]}

    def plan_cache_clear(self):
        with self.lock:
            check(lib().ssd_plan_cache_clear(self._h))

    def get_tensor(self, name):
        """Retained intermediate of the last forward as a numpy array [B,H,W,C]."""
        dims = (ctypes.c_int32 * 4)()
        cap = 1 << 20
        while True:
            buf = np.empty((cap,), np.float32)
            with self.lock:
                rc = lib().ssd_get_tensor(self._h, name.encode(),
                                          buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), cap, dims)
            if rc != 0 and b"too small" in (lib().ssd_last_error() or b"") and cap < (1 << 34):
                cap *= 8
                continue
            check(rc)
            n = dims[0] * dims[1] * dims[2] * dims[3]
            return buf[:n].reshape(dims[0], dims[1], dims[2], dims[3]).copy()

    def get_tensor_dev(self, name, shape):
        torch = _torch()
        out = torch.empty(tuple(shape), dtype=torch.float32, device="cuda:%d" % self.device)
        dims = (ctypes.c_int32 * 4)()
        with self.lock:
            check(lib().ssd_get_tensor_dev(self._h, name.encode(), _ptr(out), out.numel(), dims,
                                           _stream(torch)))
        return out

    # profiling (bench.py roofline leg)
    def profile_enable(self, on=True):
        check(lib().ssd_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        check(lib().ssd_profile_reset(self._h))

    def profile_read(self):
        names = ["conv3x3_mfma", "pointwise_mfma", "depthwise", "first_conv", "postprocess", "other",
                 "depthwise_pointwise_fused", "conv3x3_f16x3_tile256"]
        out = {}
        for i, n in enumerate(names):
            ms, cnt = ctypes.c_double(), ctypes.c_int64()
            fl, by = ctypes.c_double(), ctypes.c_double()
            check(lib().ssd_profile_read(self._h, i, ctypes.byref(ms), ctypes.byref(cnt),
                                         ctypes.byref(fl), ctypes.byref(by)))
            out[n] = {"ms": ms.value, "launches": cnt.value, "flops": fl.value, "bytes": by.value}
        return out


def network_input_size(height, width, min_dimension, divisor=128):
    """Size of the tensor the network sees for a [height, width] image, and box_scaler: the size arithmetic of
    resize_keeping_aspect_ratio (pipeline.py:138-194) -- short side -> min_dimension, long side round-half-even of
    float32(x) * float32(min_dimension / min(h, w)), padded up to a multiple of `divisor`."""
    scale = np.float32(min_dimension / min(height, width))
    if height >= width:
        nh, nw = int(np.rint(np.float32(height) * scale)), int(min_dimension)
        ph, pw = -(-nh // divisor) * divisor - nh, 0
    else:
        nw, nh = int(np.rint(np.float32(width) * scale)), int(min_dimension)
        pw, ph = -(-nw // divisor) * divisor - nw, 0
    box_scaler = np.array([nh / (nh + ph), nw / (nw + pw)] * 2, np.float32)
    return nh + ph, nw + pw, box_scaler


class RetinaNetFeatureExtractor:
    """Mirror of detector/feature_extractor.py:10-37: callable, images -> [p3, p4, p5, p6, p7].  The reference builds
    the backbone + FPN graph here; this build has ONE fused plan behind `engine`, so the call runs the engine's forward
    on `images` (uint8 [B,H,W,3] CUDA; the /255 and 2x-1 of create_pb.py:47 / mobilenet_v1.py:34 are inside the first
    kernel) and returns the retained pyramid tensors of that forward, NHWC float32 on the device."""

    def __init__(self, engine):
        self.engine = engine

    def __call__(self, images):
        B, H, W, _ = images.shape
        self.outputs = self.engine.forward(images)
        nh, nw, _bs = network_input_size(H, W, self.engine.params["min_dimension"])
        self.image_size = (nh, nw)
        return [self.engine.get_tensor_dev("p%d" % l, (B, -(-nh // s), -(-nw // s), 256)) for l, s in zip(range(3, 8), (8, 16, 32, 64, 128))]


class RetinaNetBoxPredictor:
    """Mirror of detector/box_predictor.py:11-64: callable, image_features -> {'encoded_boxes' [B,N,4],
    'class_predictions' [B,N,C]}.  The head towers ran in the same plan as the features: the call returns the retained
    head outputs of the engine's last forward (the one `RetinaNetFeatureExtractor.__call__` just ran)."""

    def __init__(self, engine, num_classes=None):
        self.engine = engine
        self.num_classes = engine.params["num_classes"] if num_classes is None else num_classes
        if self.num_classes != engine.params["num_classes"]:
            raise ValueError("num_classes differs from the engine's configuration")

    def __call__(self, image_features):
        B = image_features[0].shape[0]
        N = sum(int(f.shape[1]) * int(f.shape[2]) * 6 for f in image_features)
        return {"encoded_boxes": self.engine.get_tensor_dev("encoded_boxes", (B, N, 4)),
                "class_predictions": self.engine.get_tensor_dev("class_predictions", (B, N, self.num_classes))}


class SSD:
    """Mirror of detector/ssd.py:9-69 (inference part), with the reference's constructor:

        SSD(images, feature_extractor, anchor_generator, box_predictor, num_classes)      (ssd.py:10)

    `feature_extractor` / `box_predictor` are the two mirrors above (bound to one Engine), `anchor_generator` an
    AnchorGenerator.  As in the reference, the anchors are generated for the size of the tensor the NETWORK sees
    (ssd.py:25-31 reads it from the images tensor, which in the serving graph is already resized and padded,
    create_pb.py:44-47): for raw frames of any size that is network_input_size(H, W), not (H, W).
    Attributes: .anchors [N,4], .num_anchors_per_feature_map, .raw_predictions; .get_predictions(...) (ssd.py:42-69).
    Shorthand kept from round 1: SSD(images, engine) builds the three collaborators itself."""

    def __init__(self, images, feature_extractor, anchor_generator=None, box_predictor=None, num_classes=None):
        torch = _torch()
        if isinstance(feature_extractor, Engine):
            engine = feature_extractor
            feature_extractor = RetinaNetFeatureExtractor(engine)
            box_predictor = RetinaNetBoxPredictor(engine, num_classes)
        if anchor_generator is None:
            anchor_generator = AnchorGenerator()
        if box_predictor is None:
            raise TypeError("SSD(images, feature_extractor, anchor_generator, box_predictor, num_classes)")
        self.engine = feature_extractor.engine
        self.num_classes = self.engine.params["num_classes"] if num_classes is None else num_classes
        if self.num_classes != self.engine.params["num_classes"]:
            raise ValueError("num_classes differs from the engine's configuration")
        feature_maps = feature_extractor(images)                       # ssd.py:23
        image_height, image_width = feature_extractor.image_size      # ssd.py:25-29 (the resized, padded size)
        self.anchors = torch.from_numpy(anchor_generator(image_height, image_width)).to(images.device)     # ssd.py:31
        self.num_anchors_per_feature_map = anchor_generator.num_anchors_per_feature_map
        self.raw_predictions = box_predictor(feature_maps)            # ssd.py:37-40
        self.box_scaler = network_input_size(images.shape[1], images.shape[2], self.engine.params["min_dimension"])[2]

    def get_predictions(self, score_threshold=0.05, iou_threshold=0.5, max_boxes_per_class=20):
        """ssd.py:42-69 (+ the boxes /= box_scaler of model.py:67-68, which the reference applies right after)."""
        boxes, scores, classes, num = batch_multiclass_non_max_suppression(
            self.raw_predictions["encoded_boxes"], self.anchors,
            self.raw_predictions["class_predictions"], score_threshold=score_threshold,
            iou_threshold=iou_threshold, max_boxes_per_class=max_boxes_per_class, box_scaler=self.box_scaler)
        return {"boxes": boxes, "labels": classes, "scores": scores, "num_boxes": num}
