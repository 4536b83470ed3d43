"""Per-block phase timestamps of the 128x128 implicit-GEMM kernel (diagnostic tile id 17).
usage: python scripts/ts_phases.py H W Cin Cout k [B] [pyramid] [tile id: 17 = 128x128 (default), 18 = 64x64]
Phases (100 MHz wall clock, thread 0 of each block): start -> first stage in LDS -> K loop done ->
accumulators transposed in LDS -> last store issued."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ssd_amd
from ssd_amd._lib import check
H, W, Cin, Cout, k = [int(v) for v in sys.argv[1:6]]
B = int(sys.argv[6]) if len(sys.argv) > 6 else 32
pyr = int(sys.argv[7]) if len(sys.argv) > 7 else 0
TILE = int(sys.argv[8]) if len(sys.argv) > 8 else 17
path = "/tmp/ts_dump.bin"
os.environ["SSD_TS_DUMP"] = path
ssd_amd._lib.use_diag()        # libssd_hip_diag.so: the -DSSD_DIAG build (include/ssd_hip_diag.h)
L = ssd_amd.lib()
ms, gf = ctypes.c_double(), ctypes.c_double()
check(L.ssd_bench_conv(B, H, W, Cin, Cout, k, 1, TILE, 3, pyr, ctypes.byref(ms), ctypes.byref(gf)))
t = np.fromfile(path, dtype=np.int64).reshape(-1, 9)
st = t[:, :5].astype(np.float64) * 0.01          # us
span = st[:, 4].max() - st[:, 0].min()
print("%dx%d %d->%d k%d B=%d%s tile %s: %.3f ms/launch (events), %d blocks, span of last launch %.1f us" %
      (H, W, Cin, Cout, k, B, " pyramid" if pyr else "", "128x128" if TILE == 17 else "64x64", ms.value, len(t), span))
names = ["prologue (offsets, 2 gloads, first stage in LDS)", "K loop", "acc -> LDS transpose", "BN/act + stores issued"]
ep = t[:, 5:8].astype(np.float64) * 0.01
print("   inside the last phase: parameters arrived +%.2f us, first row stored +%.2f us, half the rows +%.2f us (means, from the transpose mark)" %
      tuple((ep[:, i] - st[:, 3]).mean() for i in range(3)))
d = np.diff(st, axis=1)
for i, n in enumerate(names):
    print("   %-52s mean %7.2f us  p10 %7.2f  p50 %7.2f  p90 %7.2f" % (n, d[:, i].mean(), *np.percentile(d[:, i], [10, 50, 90])))
life = st[:, 4] - st[:, 0]
print("   %-52s mean %7.2f us  p10 %7.2f  p50 %7.2f  p90 %7.2f" % ("block lifetime", life.mean(), *np.percentile(life, [10, 50, 90])))
print("   sum of lifetimes / span = %.1f blocks resident on average (512 slots)" % (life.sum() / span))
hw = t[:, 8] & 0xFFFFFFFF
xcc = t[:, 8] >> 32
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | ((xcc & 0xF) << 8)    # CU_ID, SE_ID(+SH), XCC
ids, cnt = np.unique(cu, return_counts=True)
print("   %d distinct (xcc, se, cu) ids; blocks per id min %d max %d" % (len(ids), cnt.min(), cnt.max()))
print("   dispatch order: blocks 0..255 sit on %d distinct CUs, blocks 256..511 on %d" % (len(np.unique(cu[:256])), len(np.unique(cu[256:512]))))
# gaps between consecutive blocks on the same slot are invisible here; idle estimate per CU:
busy = []
for i in ids[:: max(1, len(ids) // 32)]:
    s = st[cu == i]
    ev = sorted([(a, 1) for a in s[:, 0]] + [(b, -1) for b in s[:, 4]])
    depth, last, acc = 0, ev[0][0], {0: 0.0, 1: 0.0, 2: 0.0, 3: 0.0}
    for tt, dd in ev:
        acc[min(depth, 3)] += tt - last
        last = tt
        depth += dd
    tot = sum(acc.values())
    busy.append([acc[j] / tot for j in range(4)])
b = np.array(busy).mean(0)
print("   sampled CUs: time with 0 / 1 / 2 / 3+ resident blocks = %.2f / %.2f / %.2f / %.2f" % tuple(b))
# is the MFMA pipe of a CU fed?  Fraction of the launch span during which 0 / 1 / 2+ of the CU's resident blocks are inside
# their K loop (co-resident blocks that start together stay in phase: their prologues and epilogues then coincide)
t0, t1 = st[:, 0].min(), st[:, 4].max()
inloop = []
for i in ids:
    s = st[cu == i]
    ev = sorted([(a, 1) for a in s[:, 1]] + [(b_, -1) for b_ in s[:, 2]])
    depth, last, acc = 0, t0, [0.0, 0.0, 0.0]
    for tt, dd in ev:
        acc[min(depth, 2)] += tt - last
        last = tt
        depth += dd
    acc[0] += t1 - last
    inloop.append([a / (t1 - t0) for a in acc])
il = np.array(inloop).mean(0)
print("   all CUs: share of the launch span with 0 / 1 / 2+ resident blocks in their K loop = %.3f / %.3f / %.3f" % tuple(il))
# phase offset of co-resident blocks: for every block, distance of its start to the nearest start of ANOTHER block on the same CU
offs = []
for i in ids:
    s = np.sort(st[cu == i][:, 0])
    if len(s) > 1:
        dn = np.diff(s)
        offs.extend(np.minimum(np.r_[dn, np.inf], np.r_[np.inf, dn]))
offs = np.array(offs)
print("   start-to-nearest-start on the same CU: p10 %.2f  p50 %.2f  p90 %.2f us (mean lifetime %.2f)" %
      (*np.percentile(offs, [10, 50, 90]), life.mean()))
