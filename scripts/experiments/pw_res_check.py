"""pw_res.hip (1x1 convolution with the weight tile resident in LDS) against the implicit-GEMM kernel: bit-equality on a set
of shapes (option pw_res = 2 forces the kernel wherever it takes the shape), then its time on the MobileNet pointwise shapes at
a serving batch (diag library, ssd_bench_conv tile 40).
usage: python scripts/experiments/pw_res_check.py [check|bench] [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ssd_amd
what = sys.argv[1] if len(sys.argv) > 1 else "check"
if what == "check":
    from ssd_amd import ssd
    rng = np.random.default_rng(0)
    bad = 0
    for (B, H, W, Cin, Cout, bn, act) in [(2, 40, 56, 512, 512, True, "relu6"), (1, 9, 7, 32, 64, True, "relu"), (3, 13, 11, 96, 40, False, None),
                                          (2, 20, 28, 256, 1024, True, "relu6"), (1, 5, 5, 512, 24, True, None), (16, 40, 56, 512, 512, True, "relu6"),
                                          (4, 80, 112, 256, 256, False, None), (1, 1, 1, 64, 64, True, "relu")]:
        x = torch.from_numpy(rng.standard_normal((B, H, W, Cin)).astype(np.float32)).cuda()
        w = (rng.standard_normal((1, 1, Cin, Cout)) * 0.1).astype(np.float32)
        bnp = (rng.standard_normal(Cout).astype(np.float32), (1 + 0.1 * rng.standard_normal(Cout)).astype(np.float32),
               rng.standard_normal(Cout).astype(np.float32)) if bn else None
        ssd_amd.set_option("pw_res", 0)
        ref = ssd.conv2d(x, w, 1, "SAME", bn=bnp, act=act).cpu().numpy()
        ssd_amd.set_option("pw_res", 2)
        got = ssd.conv2d(x, w, 1, "SAME", bn=bnp, act=act).cpu().numpy()
        eq = np.array_equal(ref, got)
        bad += not eq
        print((B, H, W, Cin, Cout, bn, act), "bit-equal" if eq else "DIFFERENT max err %g" % np.abs(ref - got).max(), flush=True)
    sys.exit(1 if bad else 0)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ssd_amd._lib.use_diag()
L = ssd_amd.lib()
from ssd_amd._lib import check
for rnd in range(2):
    for name, H, W, Cin, Cout in (("pw 256->256 80x112", 80, 112, 256, 256), ("pw 256->512 40x56", 40, 56, 256, 512), ("pw 512->512 40x56", 40, 56, 512, 512),
                                  ("pw 512->1024 20x28", 20, 28, 512, 1024), ("pw 128->256 80x112", 80, 112, 128, 256), ("lateral4 512->256 40x56", 40, 56, 512, 256)):
        for t, tn in ((0, "128x128"), (1, "128x64"), (5, "64x64"), (40, "pw_res")):
            ms, gf = ctypes.c_double(), ctypes.c_double()
            check(L.ssd_bench_conv(B, H, W, Cin, Cout, 1, 1, t, 20, 0, ctypes.byref(ms), ctypes.byref(gf)))
            print("%-26s B=%d tile %-8s %8.3f ms  %7.1f TFLOP/s (%.1f%% of 157.3)" % (name, B, tn, ms.value, gf.value / ms.value, gf.value / ms.value / 157.3 * 100), flush=True)
