"""ctypes binding of csrc/libssd_hip.so (the C ABI of include/ssd_hip.h) and its build.

The library is built in-tree with `hipcc --offload-arch=gfx950`; there is no fallback: if it
cannot be loaded every op of this package raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SOURCES = ["api.hip", "igemm.hip", "igemm16.hip", "dwpw.hip", "elementwise.hip", "postprocess.hip"]
_LIB_PATH = os.path.join(_CSRC, "libssd_hip.so")
_lib = None

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


class SsdError(RuntimeError):
    """A non-zero return code from libssd_hip.so (message from ssd_last_error())."""


def lib_path():
    return _LIB_PATH


def build(force=False, verbose=False):
    """Compile every HIP source into csrc/libssd_hip.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, s) for s in _SOURCES]
    deps = srcs + [os.path.join(_CSRC, "ssd_internal.h"),
                   os.path.join(_HERE, "..", "include", "ssd_hip.h")]
    def fresh():
        return os.path.exists(_LIB_PATH) and \
            os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(d) for d in deps)
    if not force and fresh():
        return _LIB_PATH
    # one builder at a time (torch.distributed.run starts N ranks at once): lock, re-check,
    # compile beside the target and rename over it, so no rank ever maps a half-written file
    import fcntl
    with open(os.path.join(_CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and fresh():
                return _LIB_PATH
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            tmp = "%s.tmp.%d" % (_LIB_PATH, os.getpid())
            cmd = [hipcc] + HIPCC_FLAGS + ["-o", tmp] + srcs
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, _LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return _LIB_PATH


_f = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_vp = ctypes.c_void_p
_i = ctypes.c_int32


class SsdConfig(ctypes.Structure):
    _fields_ = [("backbone", ctypes.c_int32), ("depth_multiplier", ctypes.c_float),
                ("num_classes", ctypes.c_int32), ("score_threshold", ctypes.c_float),
                ("iou_threshold", ctypes.c_float), ("max_boxes_per_class", ctypes.c_int32),
                ("min_dimension", ctypes.c_int32), ("device", ctypes.c_int32)]


# every symbol include/ssd_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "ssd_create": (ctypes.c_int, [ctypes.POINTER(SsdConfig), ctypes.POINTER(_vp)]),
    "ssd_destroy": (None, [_vp]),
    "ssd_last_error": (ctypes.c_char_p, []),
    "ssd_load_weight": (ctypes.c_int, [_vp, ctypes.c_char_p, _f, ctypes.POINTER(ctypes.c_int64), _i]),
    "ssd_finalize": (ctypes.c_int, [_vp]),
    "ssd_forward": (ctypes.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ssd_get_tensor": (ctypes.c_int, [_vp, ctypes.c_char_p, _f, ctypes.c_int64, _i32p]),
    "ssd_get_tensor_dev": (ctypes.c_int, [_vp, ctypes.c_char_p, _vp, ctypes.c_int64, _i32p, _vp]),
    "ssd_profile_enable": (ctypes.c_int, [_vp, _i]),
    "ssd_profile_read": (ctypes.c_int, [_vp, _i, ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_int64),
                                        ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_double)]),
    "ssd_profile_reset": (ctypes.c_int, [_vp]),
    "ssd_num_anchors": (ctypes.c_int32, [_i, _i]),
    "ssd_anchors": (ctypes.c_int, [_i, _i, _f]),
    "ssd_conv2d": (ctypes.c_int, [_vp, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f,
                                  _vp, _i, _vp, _vp]),
    "ssd_conv2d_f16x3": (ctypes.c_int, [_vp, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f,
                                        _vp, _i, _vp, _vp]),
    "ssd_set_precision": (ctypes.c_int, [_vp, _i]),
    "ssd_get_precision": (ctypes.c_int, [_vp]),
    "ssd_status": (ctypes.c_int, [_vp, _i32p]),
    "ssd_depthwise3x3": (ctypes.c_int, [_vp, _i, _i, _i, _i, _f, _i, _i, _i, _i, _f, _f, _f, _i,
                                        _vp, _vp]),
    "ssd_dw_pw": (ctypes.c_int, [_vp, _i, _i, _i, _i, _f, _i, _f, _f, _f, _i, _f, _i, _f, _f, _f, _i, _vp, _vp]),
    "ssd_first_conv": (ctypes.c_int, [_vp, _i, _i, _i, _f, _i, _f, _f, _f, _i, _vp, _vp]),
    "ssd_maxpool3x3s2": (ctypes.c_int, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "ssd_concat_shuffle_split": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _i, _vp, _vp, _vp]),
    "ssd_bench_conv": (ctypes.c_int, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                      ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "ssd_bench_dwpw": (ctypes.c_int, [_i, _i, _i, _i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_double)]),
    "ssd_postprocess_workspace_bytes": (ctypes.c_size_t, [_i, _i, _i, _i]),
    "ssd_postprocess": (ctypes.c_int, [_vp, _vp, _vp, _i, _i, _i, ctypes.c_float, ctypes.c_float,
                                       _i, _f, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
}


def lib():
    """Load libssd_hip.so (building it if the sources are newer).  Raises if impossible."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().ssd_last_error()
        raise SsdError("libssd_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
