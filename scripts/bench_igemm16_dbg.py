"""Where does the time of the 256x256-tile f16x3 kernel go?  Head-tower launch (B images, 5 levels) with
pieces of the K loop removed (results wrong, timing only): SSD_IGEMM16_DBG = 0 full, 1 no DMA, 2 no DMA
and no fragment reads, 3 additionally no barrier, 4 the bare MFMA work on the 16x16x32 shape.  usage: python scripts/bench_igemm16_dbg.py [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ssd_amd
from ssd_amd._lib import check
assert torch.cuda.is_available()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ssd_amd._lib.use_diag()        # libssd_hip_diag.so: the -DSSD_DIAG build (include/ssd_hip_diag.h)
L = ssd_amd.lib()
os.environ["SSD_BENCH_PRECISION"] = "f16x3"
for rnd in range(2):
    for dbg in (0, 1, 2, 3, 4):
        os.environ["SSD_IGEMM16_DBG"] = str(dbg)
        ms, gf = ctypes.c_double(), ctypes.c_double()
        check(L.ssd_bench_conv(B, 80, 112, 256, 256, 3, 1, 0, 10, 1, ctypes.byref(ms), ctypes.byref(gf)))
        print("tower 3x3 5 levels B=%d DBG %d: %7.3f ms  %6.1f TFLOP/s algorithmic = %.3f of the f16x3 peak (838.9)"
              % (B, dbg, ms.value, gf.value / ms.value, gf.value / ms.value / 838.9), flush=True)
os.environ["SSD_IGEMM16_DBG"] = "0"
for igemm16 in ("0", "1"):
    ssd_amd.set_option("igemm16", int(igemm16))
    ms, gf = ctypes.c_double(), ctypes.c_double()
    check(L.ssd_bench_conv(B, 80, 112, 256, 256, 3, 1, 0, 10, 1, ctypes.byref(ms), ctypes.byref(gf)))
    print("option igemm16=%s: %7.3f ms %6.1f TFLOP/s" % (igemm16, ms.value, gf.value / ms.value), flush=True)
