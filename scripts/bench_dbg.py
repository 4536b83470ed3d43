import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssd_amd
from ssd_amd._lib import check
L = ssd_amd.lib()
for rnd in range(2):
    for t, name in [(0, "full"), (10, "no gload/lstore"), (11, "+no frag reads"), (12, "+no barrier")]:
        ms, gf = ctypes.c_double(), ctypes.c_double()
        check(L.ssd_bench_conv(32, 80, 112, 256, 256, 3, 1, t, 10, 1, ctypes.byref(ms), ctypes.byref(gf)))
        print("%-20s %8.3f ms %7.1f TFLOP/s (%.1f%%)" % (name, ms.value, gf.value / ms.value, gf.value / ms.value / 1.573), flush=True)
