"""One conv shape, few launches: the target of rocprofv3 --pmc passes.
usage: python scripts/bench_one.py tile [reps] [shape]   shape: tower | pw512 | logits | pw2 | pw3 | pw5"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ssd_amd
from ssd_amd._lib import check
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
shape = sys.argv[3] if len(sys.argv) > 3 else "tower"
ssd_amd._lib.use_diag()        # libssd_hip_diag.so: the -DSSD_DIAG build (include/ssd_hip_diag.h)
L = ssd_amd.lib()
ms, gf = ctypes.c_double(), ctypes.c_double()
if shape == "tower":
    check(L.ssd_bench_conv(32, 80, 112, 256, 256, 3, 1, tile, reps, 1, ctypes.byref(ms), ctypes.byref(gf)))
elif shape == "logits":
    check(L.ssd_bench_conv(32, 80, 112, 256, 480, 3, 1, tile, reps, 1, ctypes.byref(ms), ctypes.byref(gf)))
elif shape == "pw2":
    check(L.ssd_bench_conv(32, 160, 224, 64, 128, 1, 1, tile, reps, 0, ctypes.byref(ms), ctypes.byref(gf)))
elif shape == "pw3":
    check(L.ssd_bench_conv(32, 160, 224, 128, 128, 1, 1, tile, reps, 0, ctypes.byref(ms), ctypes.byref(gf)))
elif shape == "pw5":
    check(L.ssd_bench_conv(32, 80, 112, 256, 256, 1, 1, tile, reps, 0, ctypes.byref(ms), ctypes.byref(gf)))
else:
    check(L.ssd_bench_conv(32, 40, 56, 512, 512, 1, 1, tile, reps, 0, ctypes.byref(ms), ctypes.byref(gf)))
print("%s tile %d: %.3f ms %.1f TFLOP/s" % (shape, tile, ms.value, gf.value / ms.value))
