"""A/B the implicit-GEMM tile variants on the layer shapes of the detector (one process).
usage: python scripts/bench_conv.py [B]"""
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
TILES = {8: "128x32 deep", 0: "128x128", 1: "128x64", 2: "128x32", 3: "128x256", 4: "256x128", 5: "64x64", 6: "128x96",
         20: "lat 1x1", 21: "lat 1x2", 22: "lat 2x1", 23: "lat 2x2", 24: "lat w2 1x1", 25: "lat w4 1x1", 26: "lat w4 2x1",
         27: "lat w4 1x2", 28: "lat il", 29: "lat il D8", 30: "lat il D16", 31: "lat il nm", 32: "lat il D8 nm", 33: "abl no lds", 34: "abl no lds,x", 35: "abl no lds,x,w", 36: "abl mfma only", 7: "64x64 deep", -1: "auto"}


def run(name, H, W, Cin, Cout, k, stride, tiles, pyramid=0, reps=10):
    for t in tiles:
        ms, gf = ctypes.c_double(), ctypes.c_double()
        check(L.ssd_bench_conv(B, H, W, Cin, Cout, k, stride, t, reps, pyramid, ctypes.byref(ms), ctypes.byref(gf)))
        print("%-34s tile %-8s %8.3f ms  %7.1f TFLOP/s (%.1f%% of 157.3)" %
              (name, TILES[t], ms.value, gf.value / ms.value, gf.value / ms.value / 157.3 * 100), flush=True)


if len(sys.argv) > 2 and sys.argv[2] == "box":       # the box head alone: 128x32 tiles with one / two register sets of operand loads
    for rnd in range(3):
        run("boxes 3x3 256->24 5 levels", 80, 112, 256, 24, 3, 1, [2, 8], pyramid=1)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "p67":       # fpn p6 / p7 at a serving batch: 64x64 tiles against the block forms of the latency kernel
    for rnd in range(2):
        for shape in (("fpn p6 3x3 s2 1024->256 20x28", 20, 28, 1024, 256, 3, 2), ("fpn p7 3x3 s2 256->256 10x14", 10, 14, 256, 256, 3, 2)):
            run(*shape, [5, 7, 24, 25, 26, 27, 20])
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "lat1":      # the one-wave latency form alone: prefetch depth x tile order
    for rnd in range(2):
        for shape in (("fpn p6 3x3 s2 1024->256 20x28", 20, 28, 1024, 256, 3, 2), ("fpn p7 3x3 s2 256->256 10x14", 10, 14, 256, 256, 3, 2),
                      ("lateral5 1x1 1024->256 20x28", 20, 28, 1024, 256, 1, 1)):
            run(*shape, [20, 30, 33, 34, 35, 36])
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "sn":        # ShuffleNet's conv1x1_before layers (K = 58 / 116 / 232: 2 / 4 / 8 K-steps) at the serving batch
    for rnd in range(2):
        for shape in (("sn 58->58 80x80", 80, 80, 58, 58, 1, 1), ("sn 116->116 40x40", 40, 40, 116, 116, 1, 1),
                      ("sn 232->232 20x20", 20, 20, 232, 232, 1, 1), ("sn 116->116 80x80", 80, 80, 116, 116, 1, 1),
                      ("sn 232->232 40x40", 40, 40, 232, 232, 1, 1)):
            run(*shape, [-1, 0, 1, 5, 25, 26, 27])
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "pw":        # the backbone's 1x1 layers at a serving batch: 32x32x2 tiles against the four-wave latency form
    for rnd in range(2):
        for shape in (("pw 256->256 80x112", 80, 112, 256, 256, 1, 1), ("pw 256->512 40x56", 40, 56, 256, 512, 1, 1),
                      ("pw 512->512 40x56", 40, 56, 512, 512, 1, 1), ("pw 512->1024 20x28", 20, 28, 512, 1024, 1, 1),
                      ("pw 1024->1024 20x28", 20, 28, 1024, 1024, 1, 1)):
            run(*shape, [0, 1, 5, 7, 25, 26, 27])
    sys.exit(0)
for rnd in range(2):
    run("tower 3x3 256->256 5 levels", 80, 112, 256, 256, 3, 1, [0, 1, 5], pyramid=1)
run("fpn p3 3x3 256->256 80x112", 80, 112, 256, 256, 3, 1, [0])
run("logits 3x3 256->480 5 levels", 80, 112, 256, 480, 3, 1, [0, 6], pyramid=1)
run("boxes 3x3 256->24 5 levels", 80, 112, 256, 24, 3, 1, [2], pyramid=1)
run("pw 32->64 320x448", 320, 448, 32, 64, 1, 1, [1, 0])
run("pw 64->128 160x224", 160, 224, 64, 128, 1, 1, [0, 1, 5])
run("pw 128->128 160x224", 160, 224, 128, 128, 1, 1, [0, 1, 5])
run("pw 128->256 80x112", 80, 112, 128, 256, 1, 1, [0, 1, 5])
run("pw 256->256 80x112", 80, 112, 256, 256, 1, 1, [0, 1, 5])
run("pw 256->512 40x56", 40, 56, 256, 512, 1, 1, [0, 1, 5])
run("pw 512->512 40x56", 40, 56, 512, 512, 1, 1, [0, 1, 5])
run("pw 512->1024 20x28", 20, 28, 512, 1024, 1, 1, [0, 1, 5])
run("pw 1024->1024 20x28", 20, 28, 1024, 1024, 1, 1, [0, 1, 5])

# the latency form (igemm_lat.hip: 16x16x4 MFMA, one wave per block) against the 64x64-tile kernel on the small launches of a batch-1 / batch-2 forward
if B <= 4:
    LAT = [5, 7, 20, 21, 22, 23, 24, 25, 26, 27, -1]
    run("boxes 3x3 256->24 5 levels", 80, 112, 256, 24, 3, 1, [2, 20, 21, 22, 23, 24], pyramid=1)
    run("fpn p6 3x3 s2 1024->256 20x28", 20, 28, 1024, 256, 3, 2, LAT)
    run("fpn p7 3x3 s2 256->256 10x14", 10, 14, 256, 256, 3, 2, LAT)
    run("fpn p5 3x3 256->256 20x28", 20, 28, 256, 256, 3, 1, LAT)
    run("fpn p4 3x3 256->256 40x56", 40, 56, 256, 256, 3, 1, LAT)
    run("fpn p3 3x3 256->256 80x112", 80, 112, 256, 256, 3, 1, LAT)
    run("lateral5 1x1 1024->256 20x28", 20, 28, 1024, 256, 1, 1, LAT)
    run("lateral4 1x1 512->256 40x56", 40, 56, 512, 256, 1, 1, LAT)
    run("lateral3 1x1 256->256 80x112", 80, 112, 256, 256, 1, 1, LAT)
    run("pw 256->512 40x56", 40, 56, 256, 512, 1, 1, LAT)
    run("pw 512->512 40x56", 40, 56, 512, 512, 1, 1, LAT)
    run("pw 512->1024 20x28", 20, 28, 512, 1024, 1, 1, LAT)
    run("pw 1024->1024 20x28", 20, 28, 1024, 1024, 1, 1, LAT)
    run("pw 256->256 80x112", 80, 112, 256, 256, 1, 1, LAT)
    run("tower 3x3 256->256 5 levels", 80, 112, 256, 256, 3, 1, [5, 7, 23, 21, 25, 26, 27], pyramid=1)
    run("logits 3x3 256->480 5 levels", 80, 112, 256, 480, 3, 1, [6, 23], pyramid=1)
