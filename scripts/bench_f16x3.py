"""f32 vs f16x3 implicit-GEMM on the head / FPN / pointwise shapes (one process).
usage: python scripts/bench_f16x3.py [B]"""
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
TILES = {-1: "auto", 0: "128x128", 1: "128x64", 2: "128x32", 5: "64x64", 6: "128x96"}


def run(name, H, W, Cin, Cout, k, stride, tiles, pyramid=0, reps=10):
    for prec in ("f32", "f16x3"):
        os.environ["SSD_BENCH_PRECISION"] = prec
        for t in tiles:
            ms, gf = ctypes.c_double(), ctypes.c_double()
            check(L.ssd_bench_conv(B, H, W, Cin, Cout, k, stride, t, reps, pyramid, ctypes.byref(ms), ctypes.byref(gf)))
            print("%-30s %-6s tile %-8s %8.3f ms  %7.1f TFLOP/s algorithmic" %
                  (name, prec, TILES[t], ms.value, gf.value / ms.value), flush=True)


for rnd in range(2):
    run("tower 3x3 256->256 5 levels", 80, 112, 256, 256, 3, 1, [0, 1], pyramid=1)
run("fpn p3 3x3 256->256 80x112", 80, 112, 256, 256, 3, 1, [0])
run("logits 3x3 256->480 5 levels", 80, 112, 256, 480, 3, 1, [0, 6], pyramid=1)
run("boxes 3x3 256->24 5 levels", 80, 112, 256, 24, 3, 1, [2], pyramid=1)
run("pw 256->256 80x112", 80, 112, 256, 256, 1, 1, [0])
run("pw 512->512 40x56", 40, 56, 512, 512, 1, 1, [0])
run("pw 1024->1024 20x28", 20, 28, 1024, 1024, 1, 1, [0])
