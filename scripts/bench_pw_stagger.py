"""1x1 implicit-GEMM launches with the first-round blocks of a CU's second (third ...) slot started late
(IgemmArgs.stagger_step, SSD_PW_STAGGER in 10-ns ticks): the co-resident blocks otherwise stay in phase and their
prologues / epilogues coincide (scripts/ts_phases.py).   usage: python scripts/bench_pw_stagger.py [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ssd_amd
from ssd_amd._lib import check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ssd_amd._lib.use_diag()
L = ssd_amd.lib()
TILES = {0: "128x128", 1: "128x64", 5: "64x64"}
SHAPES = [("pw 256->256 80x112", 80, 112, 256, 256), ("pw 256->512 40x56", 40, 56, 256, 512), ("pw 512->512 40x56", 40, 56, 512, 512),
          ("pw 512->1024 20x28", 20, 28, 512, 1024), ("pw 1024->1024 20x28", 20, 28, 1024, 1024)]
for name, H, W, Cin, Cout in SHAPES:
    for t in (0, 1, 5):
        row = []
        for st in (0, 500, 1000, 1500, 2000, 3000):
            os.environ["SSD_PW_STAGGER"] = str(st)
            ms, gf = ctypes.c_double(), ctypes.c_double()
            check(L.ssd_bench_conv(B, H, W, Cin, Cout, 1, 1, t, 20, 0, ctypes.byref(ms), ctypes.byref(gf)))
            row.append("%d: %.1f us %.1f%%" % (st, ms.value * 1e3, gf.value / ms.value / 157.3 * 100))
        print("%-22s B=%d tile %-8s  %s" % (name, B, TILES[t], " | ".join(row)), flush=True)
