"""ctypes binding of csrc/libssd_hip.so (the C ABI of include/ssd_hip.h) and its build.

The library is built in-tree with `hipcc --offload-arch=gfx950`; there is no fallback: if it
cannot be loaded every op of this package raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SOURCES = ["abi.hip", "weights.hip", "plan.hip", "stages.hip", "igemm.hip", "igemm_lat.hip", "igemm16.hip", "dwpw_stream.hip", "sn_pw.hip", "front.hip",
            "elementwise.hip", "postprocess.hip"]
_LIB_PATH = os.path.join(_CSRC, "libssd_hip.so")
_DIAG_PATH = os.path.join(_CSRC, "libssd_hip_diag.so")       # -DSSD_DIAG build, scripts/ only
_lib = None
_diag = False

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


class SsdError(RuntimeError):
    """A non-zero return code from libssd_hip.so (message from ssd_last_error())."""


def lib_path():
    return _LIB_PATH


def build(force=False, verbose=False, diag=False):
    """Compile every HIP source into csrc/libssd_hip.so (cross-compiles without a GPU); returns early
    when the library is newer than every source.  diag=True builds csrc/libssd_hip_diag.so instead:
    the same sources with -DSSD_DIAG (ablation kernels, tile overrides, phase stamps, ssd_bench_*;
    include/ssd_hip_diag.h) for scripts/ -- never loaded by the product."""
    srcs = [os.path.join(_CSRC, s) for s in _SOURCES]
    deps = srcs + [os.path.join(_CSRC, "ssd_internal.h"), os.path.join(_CSRC, "host.h"), os.path.join(_CSRC, "igemm_mfma16.h"),
                   os.path.join(_HERE, "..", "include", "ssd_hip.h"),
                   os.path.join(_HERE, "..", "include", "ssd_hip_diag.h"), os.path.join(_CSRC, "exports.map")]
    target = _DIAG_PATH if diag else _LIB_PATH
    # freshness by CONTENT (sha256 of sources + headers + flags, kept beside the library), not by mtime: the
    # snapshot that travels to a GPU box does not promise to preserve modification-time order
    import hashlib
    hh = hashlib.sha256(" ".join(HIPCC_FLAGS + (["-DSSD_DIAG"] if diag else [])).encode())
    for d in deps:
        with open(d, "rb") as f:
            hh.update(f.read())
    digest = hh.hexdigest()
    stamp = target + ".sha"
    def fresh():
        if not (os.path.exists(target) and os.path.exists(stamp)):
            return False
        with open(stamp) as f:
            return f.read().strip() == digest
    if not force and fresh():
        return target
    # one builder at a time (torch.distributed.run starts N ranks at once): lock, re-check,
    # compile beside the target and rename over it, so no rank ever maps a half-written file
    import fcntl
    with open(os.path.join(_CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and fresh():
                return target
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            tmp = "%s.tmp.%d" % (target, os.getpid())
            # one object per source, compiled in parallel and kept (build/ is git-ignored): an edit to one
            # kernel file recompiles that file only
            objdir = os.path.join(_CSRC, "build", "diag" if diag else "ship")
            os.makedirs(objdir, exist_ok=True)
            hdrs = deps[len(srcs):]
            cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + (["-DSSD_DIAG"] if diag else [])
            jobs = []
            for src in srcs:
                obj = os.path.join(objdir, os.path.basename(src) + ".o")
                oh = hashlib.sha256(" ".join(cflags).encode())
                for d in [src] + hdrs:
                    with open(d, "rb") as f:
                        oh.update(f.read())
                ostamp, odig = obj + ".sha", oh.hexdigest()
                if force or not os.path.exists(obj) or not os.path.exists(ostamp) or open(ostamp).read().strip() != odig:
                    jobs.append(([hipcc] + cflags + ["-c", src, "-o", obj], ostamp, odig))
            if verbose:
                for j in jobs:
                    print(" ".join(j[0]))
            procs = [subprocess.Popen(j[0]) for j in jobs]
            rcs = [p.wait() for p in procs]
            if any(rcs):
                raise subprocess.CalledProcessError(max(rcs), "hipcc -c")
            for _cmd, ostamp, odig in jobs:
                with open(ostamp, "w") as f:
                    f.write(odig)
            objs = [os.path.join(objdir, os.path.basename(src) + ".o") for src in srcs]
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(_CSRC, "exports.map"),
                   "-o", tmp] + objs
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            # hipcc 7.2 can drop a kernel's host stub without a diagnostic (csrc/dwpw_stream.hip, note in dma_b): an
            # undefined symbol must fail the BUILD, not the first import on the GPU box
            ctypes.CDLL(tmp)
            os.replace(tmp, target)
            with open(stamp, "w") as f:
                f.write(digest)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return target


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
    "ssd_record_words": (ctypes.c_int32, [_vp]),
    "ssd_forward_records": (ctypes.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ssd_forward_host": (ctypes.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ssd_network_shape": (ctypes.c_int, [_vp, _i, _i, _i32p]),
    "ssd_forward_mixed": (ctypes.c_int, [_vp, _vp, _i, _i32p, ctypes.POINTER(ctypes.c_int64), _vp, _vp]),
    "ssd_forward_mixed_host": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_void_p), _i, _i32p, _vp, _vp]),
    "ssd_detect_host": (ctypes.c_int, [_vp, _vp, _i, _i, ctypes.c_float, _vp, _vp, _vp, _vp, _i, _i32p, _vp]),
    "ssd_get_tensor": (ctypes.c_int, [_vp, ctypes.c_char_p, _f, ctypes.c_int64, _i32p]),
    "ssd_get_tensor_dev": (ctypes.c_int, [_vp, ctypes.c_char_p, _vp, ctypes.c_int64, _i32p, _vp]),
    "ssd_plan_cache_stats": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    "ssd_plan_cache_clear": (ctypes.c_int, [_vp]),
    "ssd_profile_enable": (ctypes.c_int, [_vp, _i]),
    "ssd_profile_read": (ctypes.c_int, [_vp, _i, ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_int64),
                                        ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_double)]),
    "ssd_profile_reset": (ctypes.c_int, [_vp]),
    "ssd_num_anchors": (ctypes.c_int32, [_i, _i]),
    "ssd_anchors": (ctypes.c_int, [_i, _i, _f]),
    "ssd_anchors_ex": (ctypes.c_int64, [_i, _i, _i, _i32p, ctypes.POINTER(ctypes.c_double), _i, ctypes.POINTER(ctypes.c_double),
                                        _i, ctypes.POINTER(ctypes.c_double), _f, ctypes.c_int64]),
    "ssd_conv2d": (ctypes.c_int, [_vp, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f,
                                  _vp, _i, _vp, _vp]),
    "ssd_conv2d_f16x3": (ctypes.c_int, [_vp, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f,
                                        _vp, _i, _vp, _vp]),
    "ssd_set_option": (ctypes.c_int, [_vp, ctypes.c_char_p, _i]),
    "ssd_get_option": (ctypes.c_int, [_vp, ctypes.c_char_p, _i32p]),
    "ssd_set_precision": (ctypes.c_int, [_vp, _i]),
    "ssd_get_precision": (ctypes.c_int, [_vp]),
    "ssd_status": (ctypes.c_int, [_vp, _i32p]),
    "ssd_depthwise3x3": (ctypes.c_int, [_vp, _i, _i, _i, _i, _f, _i, _i, _i, _i, _f, _f, _f, _i,
                                        _vp, _vp]),
    "ssd_dw_pw": (ctypes.c_int, [_vp, _i, _i, _i, _i, _f, _i, _f, _f, _f, _i, _f, _i, _f, _f, _f, _i, _vp, _vp]),
    "ssd_first_conv": (ctypes.c_int, [_vp, _i, _i, _i, _f, _i, _f, _f, _f, _i, _vp, _vp]),
    "ssd_front_block": (ctypes.c_int, [_vp, _i, _i, _i, _f, _i, _f, _f, _f, _i, _f, _f, _f, _f, _i, _f, _i, _f, _f, _f, _i, _vp, _vp]),
    "ssd_first_conv_maxpool": (ctypes.c_int, [_vp, _i, _i, _i, _f, _i, _f, _f, _f, _i, _vp, _vp]),
    "ssd_maxpool3x3s2": (ctypes.c_int, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "ssd_concat_shuffle_split": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _i, _vp, _vp, _vp]),
    "ssd_shuffle_conv1x1": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _i, _f, _i, _f, _f, _f, _i, _vp, _vp]),
    "ssd_postprocess_workspace_bytes": (ctypes.c_size_t, [_i, _i, _i, _i]),
    "ssd_postprocess": (ctypes.c_int, [_vp, _vp, _vp, _i, _i, _i, ctypes.c_float, ctypes.c_float,
                                       _i, _f, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
}


# include/ssd_hip_diag.h (libssd_hip_diag.so only)
DIAG_SIGNATURES = {
    "ssd_bench_conv": (ctypes.c_int, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                      ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "ssd_bench_dwpw": (ctypes.c_int, [_i, _i, _i, _i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_double)]),
}


def use_diag():
    """scripts/ only: make lib() load the -DSSD_DIAG build.  Must be called before the first lib()."""
    global _diag
    if _lib is not None and not _diag:
        raise RuntimeError("use_diag() after libssd_hip.so was loaded")
    _diag = True


def lib():
    """Load libssd_hip.so, rebuilding it first when any source is newer (build() returns at once when
    it is fresh; a source tree without hipcc and with a fresh library -- the GPU box -- never compiles).
    Raises if impossible: there is no fallback."""
    global _lib
    if _lib is None:
        target = _DIAG_PATH if _diag else _LIB_PATH
        try:
            path = build(diag=_diag)
        except (FileNotFoundError, subprocess.CalledProcessError, OSError) as e:
            # a deployment that ships the package with a prebuilt library and no sources / headers / hipcc: its
            # freshness cannot be checked -- load it and say so; only a missing library is fatal
            if not os.path.exists(target):
                raise RuntimeError("single-shot-detector_amd: %s is missing and cannot be built here (%r); there is no "
                                   "fallback" % (os.path.basename(target), e)) from e
            import warnings
            warnings.warn("single-shot-detector_amd: loading the existing %s without a freshness check (%r)"
                          % (os.path.basename(target), e))
            path = target
        _lib = ctypes.CDLL(path)
        sigs = dict(SIGNATURES, **DIAG_SIGNATURES) if _diag else SIGNATURES
        for name, (res, args) in sigs.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def set_option(key, value, handle=None):
    """ssd_set_option: a kernel / schedule selector of the library (include/ssd_hip.h lists the keys), process-wide when
    `handle` is None.  None of them changes a result bit in mode f32."""
    check(lib().ssd_set_option(handle, key.encode(), int(value)))


def get_option(key, handle=None):
    v = ctypes.c_int32()
    check(lib().ssd_get_option(handle, key.encode(), ctypes.byref(v)))
    return v.value


def check(rc):
    if rc != 0:
        msg = lib().ssd_last_error()
        raise SsdError("libssd_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
