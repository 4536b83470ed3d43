"""Implicit-GEMM launches in the batch-norm form with the LDS-transposed epilogue (0) and with transposed accumulators
(1, IgemmArgs.tr / SSD_IGEMM_TR).   usage: python scripts/bench_tr.py [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ssd_amd
from ssd_amd._lib import check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ssd_amd._lib.use_diag()
L = ssd_amd.lib()
TILES = {0: "128x128", 1: "128x64", 5: "64x64"}
SHAPES = [("tower 3x3 256->256 5 levels", 80, 112, 256, 256, 3, 1, (0,)), ("fpn p3 3x3 256->256 80x112", 80, 112, 256, 256, 3, 0, (0,)),
          ("pw 128->256 80x112", 80, 112, 128, 256, 1, 0, (0, 1, 5)), ("pw 256->256 80x112", 80, 112, 256, 256, 1, 0, (0, 1, 5)),
          ("pw 256->512 40x56", 40, 56, 256, 512, 1, 0, (0, 1, 5)), ("pw 512->512 40x56", 40, 56, 512, 512, 1, 0, (0, 1, 5)),
          ("pw 512->1024 20x28", 20, 28, 512, 1024, 1, 0, (0, 1, 5)), ("pw 1024->1024 20x28", 20, 28, 1024, 1024, 1, 0, (0, 1, 5))]
for rnd in range(2):
    for name, H, W, Cin, Cout, k, pyr, tiles in SHAPES:
        for t in tiles:
            row = []
            for tr in (0, 1):
                os.environ["SSD_IGEMM_TR"] = str(tr)
                ms, gf = ctypes.c_double(), ctypes.c_double()
                check(L.ssd_bench_conv(B, H, W, Cin, Cout, k, 1, t, 10 if k == 3 else 20, pyr, ctypes.byref(ms), ctypes.byref(gf)))
                row.append("tr=%d %8.1f us %5.1f%%" % (tr, ms.value * 1e3, gf.value / ms.value / 157.3 * 100))
            print("%-28s B=%d tile %-8s  %s" % (name, B, TILES[t], " | ".join(row)), flush=True)
