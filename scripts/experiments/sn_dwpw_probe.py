"""ShuffleNet-shaped depthwise + 1x1 pairs on dwpw_stream.hip in its DENSE form (one contiguous output tensor), with the
in-kernel phase cycle totals of the diag build: what the units of a stage cost without their destination maps.
usage (GPU box): python scripts/experiments/sn_dwpw_probe.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, ssd_amd
from ssd_amd._lib import check
ssd_amd._lib.use_diag()
L = ssd_amd.lib()
path = "/tmp/ts_dwpw.bin"
names = ["wait", "barrier1", "B issue + depthwise", "barrier2", "DMA issue", "MFMA", "epilogue"]
for B in (32, 64):
    for (H, W, C, Co, s) in [(80, 80, 58, 58, 1), (40, 40, 116, 116, 1), (20, 20, 232, 232, 1), (160, 160, 24, 58, 2), (80, 80, 64, 64, 1), (320, 448, 32, 64, 1)]:
        if (H, W) == (320, 448) and B == 64: continue
        ms = ctypes.c_double()
        check(L.ssd_bench_dwpw(B, H, W, C, Co, s, 0, 10, ctypes.byref(ms))); pair = ms.value
        os.environ["SSD_TS_DUMP"] = path
        check(L.ssd_bench_dwpw(B, H, W, C, Co, s, 1, 10, ctypes.byref(ms))); os.environ.pop("SSD_TS_DUMP")
        ph = np.fromfile(path, dtype=np.int64).reshape(-1, 8).astype(np.float64); ph = ph[ph[:, 7] > 0]
        per_it = ph[:, :7].sum(0) / ph[:, 7].sum()
        mb = B * (H * W * C + (H // s) * (W // s) * Co) * 4 / 1e6      # MB moved
        print("B=%d %dx%d %d->%d s%d: pair %.1f us, fused %.1f us (%.2f TB/s), iterations/block %.1f, cycles/iteration %.0f: %s" % (
            B, H, W, C, Co, s, pair * 1e3, ms.value * 1e3, mb / ms.value / 1e3, ph[:, 7].mean(), per_it.sum(),
            " ".join("%s %.0f" % (n, v) for n, v in zip(names, per_it))), flush=True)
