import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssd_amd
from ssd_amd._lib import check
L = ssd_amd.lib()
shapes = {"pw512 40x56": (32, 40, 56, 512, 512, 1, 1, 0), "pw256 80x112": (32, 80, 112, 256, 256, 1, 1, 0), "tower": (32, 80, 112, 256, 256, 3, 1, 1)}
for nm, (B, H, W, Ci, Co, k, st, pyr) in shapes.items():
    for rnd in range(2):
        for t, name in [(0, "full"), (15, "quarter of stores"), (13, "no global stores")]:
            ms, gf = ctypes.c_double(), ctypes.c_double()
            check(L.ssd_bench_conv(B, H, W, Ci, Co, k, st, t, 20, pyr, ctypes.byref(ms), ctypes.byref(gf)))
            if rnd == 1:
                print("%-14s %-20s %8.3f ms %7.1f TFLOP/s (%.1f%%)" % (nm, name, ms.value, gf.value / ms.value, gf.value / ms.value / 1.573), flush=True)
