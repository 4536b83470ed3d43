import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssd_amd
from ssd_amd._lib import check
L = ssd_amd.lib()
shapes = {"pw512 40x56": (32, 40, 56, 512, 512, 1, 1, 0), "pw256 80x112": (32, 80, 112, 256, 256, 1, 1, 0), "pw128 160x224": (32, 160, 224, 128, 128, 1, 1, 0),
          "tower": (32, 80, 112, 256, 256, 3, 1, 1), "logits": (32, 80, 112, 256, 480, 3, 1, 1)}
for nm, (B, H, W, Ci, Co, k, st, pyr) in shapes.items():
    ms, gf = ctypes.c_double(), ctypes.c_double()
    for rep in range(2):
        check(L.ssd_bench_conv(B, H, W, Ci, Co, k, st, -1, 20, pyr, ctypes.byref(ms), ctypes.byref(gf)))
    print("%-14s %8.3f ms %7.1f TFLOP/s (%.1f%%)" % (nm, ms.value, gf.value / ms.value, gf.value / ms.value / 1.573), flush=True)
