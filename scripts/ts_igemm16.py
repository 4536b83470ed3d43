"""Per-block phase timestamps of the 256x256-tile f16x3 kernel (igemm16.hip, diagnostic build DBG 7).
usage: python scripts/ts_igemm16.py [B] [Cin]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ssd_amd
from ssd_amd._lib import check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Cin = int(sys.argv[2]) if len(sys.argv) > 2 else 256
path = "/tmp/ts16.bin"
os.environ["SSD_TS_DUMP"] = path
os.environ["SSD_BENCH_PRECISION"] = "f16x3"
ssd_amd._lib.use_diag()        # libssd_hip_diag.so: the -DSSD_DIAG build (include/ssd_hip_diag.h)
L = ssd_amd.lib()
ssd_amd.set_option("igemm16", 1)
ms, gf = ctypes.c_double(), ctypes.c_double()
check(L.ssd_bench_conv(B, 80, 112, Cin, 256, 3, 1, 17, 3, 1, ctypes.byref(ms), ctypes.byref(gf)))
t = np.fromfile(path, dtype=np.int64).reshape(-1, 9)
t = t[t[:, 0] != 0]
st = t[:, :5].astype(np.float64) * 0.01
span = st[:, 4].max() - st[:, 0].min()
print("tower 3x3 %d->256 B=%d: %.3f ms/launch, %d blocks, span %.1f us" % (Cin, B, ms.value, len(t), span))
names = ["prologue (offsets, first stage landed)", "K loop", "epilogue until last store issued", "stores retired"]
d = np.diff(st, axis=1)
for i, n in enumerate(names):
    print("   %-42s mean %8.2f us  p10 %8.2f  p50 %8.2f  p90 %8.2f" % (n, d[:, i].mean(), *np.percentile(d[:, i], [10, 50, 90])))
ep = t[:, 5:8].astype(np.float64) * 0.01
print("   inside the epilogue, from the end of the K loop: parameters arrived +%.2f us, first transpose in LDS +%.2f us, first of 4 passes stored +%.2f us"
      % tuple((ep[:, i] - st[:, 2]).mean() for i in range(3)))
life = st[:, 4] - st[:, 0]
print("   %-42s mean %8.2f us  p10 %8.2f  p50 %8.2f  p90 %8.2f" % ("block lifetime", life.mean(), *np.percentile(life, [10, 50, 90])))
print("   sum of lifetimes / span = %.1f blocks resident on average (256 CUs)" % (life.sum() / span))
ks = 9 * Cin // 32
print("   K loop per K-step: %.3f us (3072 matrix cycles = 1.28 us at 2.4 GHz)" % (d[:, 1].mean() / ks))
# start-time gaps between consecutive blocks on one CU
hw = t[:, 8] & 0xFFFFFFFF
xcc = t[:, 8] >> 32
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | ((xcc & 0xF) << 8)
gaps = []
for i in np.unique(cu):
    s = st[cu == i]
    s = s[np.argsort(s[:, 0])]
    gaps += list(s[1:, 0] - s[:-1, 4])
gaps = np.array(gaps)
print("   gap between a block's end and the next block's start on the same CU: mean %.2f us p50 %.2f p90 %.2f (n=%d)" %
      (gaps.mean(), *np.percentile(gaps, [50, 90]), len(gaps)))
