"""Diagnostic: what the last, partly filled round of tiles costs a LONE 3x3 launch.  fpn p3's shape (80 x 112, 256 -> 256, 128x128
tiles: 140 blocks per image, 512 block slots on the chip) at batch sizes around 32: TFLOP/s against rounds = blocks / 512.
usage: python scripts/experiments/tail_rounds.py        (diag library)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ssd_amd
from ssd_amd._lib import check
ssd_amd._lib.use_diag()
L = ssd_amd.lib()
for rnd in range(2):
    for B in (22, 25, 26, 29, 30, 32, 33, 36, 37, 40, 44, 51, 52, 55):
        ms, gf = ctypes.c_double(), ctypes.c_double()
        for _ in range(2):
            check(L.ssd_bench_conv(B, 80, 112, 256, 256, 3, 1, 0, 10, 0, ctypes.byref(ms), ctypes.byref(gf)))
        blocks = 140 * B
        print("B %2d  blocks %5d  rounds %6.2f  %7.3f ms  %6.1f TFLOP/s  %.3f of peak" % (B, blocks, blocks / 512.0, ms.value, gf.value / ms.value, gf.value / ms.value / 157.3), flush=True)
