"""Timing ablations of the streaming depthwise+pointwise kernel (libssd_hip_diag.so; results of ablated runs are wrong).
usage: python scripts/abl_dwpws.py [B] [layers]   mask bits: 1 no patch DMA in the loop, 2 no B / weight DMA, 4 no depthwise
FMAs, 8 no MFMA, 16 no stores"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssd_amd
from ssd_amd._lib import check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sel = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 3, 7]
ssd_amd._lib.use_diag()
L = ssd_amd.lib()
LAYERS = {1: (320, 448, 32, 64, 1), 2: (320, 448, 64, 128, 2), 3: (160, 224, 128, 128, 1), 4: (160, 224, 128, 256, 2),
          5: (80, 112, 256, 256, 1), 6: (80, 112, 256, 512, 2), 7: (40, 56, 512, 512, 1), 12: (40, 56, 512, 1024, 2),
          13: (20, 28, 1024, 1024, 1)}
for i in sel:
    H, W, C, Co, s = LAYERS[i]
    out = []
    for mask in [int(v) for v in os.environ.get('ABL_MASKS', '0,1,2,3,4,8,16,19,23,31').split(',')]:
        os.environ["SSD_DWPWS_ABL"] = str(mask)
        ms = ctypes.c_double()
        check(L.ssd_bench_dwpw(B, H, W, C, Co, s, 1, 10, ctypes.byref(ms)))
        out.append("%d: %.3f" % (mask, ms.value))
    print("Conv2d_%d B=%d  ms by ablation mask  " % (i, B) + "  ".join(out), flush=True)
