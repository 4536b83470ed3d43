"""Depthwise+pointwise blocks of MobileNet-v1 at 640x896: the streaming fused kernel (dwpw_stream.hip) vs the two-kernel
pair, and the fused kernel's phase cycle totals.  usage: python scripts/bench_dwpw.py [B] [layers]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ssd_amd
from ssd_amd._lib import check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sel = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4, 5, 6]
ssd_amd._lib.use_diag()        # libssd_hip_diag.so: the -DSSD_DIAG build (include/ssd_hip_diag.h)
L = ssd_amd.lib()
# layer: (H, W of the depthwise input, C, Cout, stride)
LAYERS = {1: (320, 448, 32, 64, 1), 2: (320, 448, 64, 128, 2), 3: (160, 224, 128, 128, 1), 4: (160, 224, 128, 256, 2),
          5: (80, 112, 256, 256, 1), 6: (80, 112, 256, 512, 2), 7: (40, 56, 512, 512, 1), 12: (40, 56, 512, 1024, 2),
          13: (20, 28, 1024, 1024, 1)}
path = "/tmp/ts_dwpw.bin"
for i in sel:
    H, W, C, Co, s = LAYERS[i]
    ms = ctypes.c_double()
    check(L.ssd_bench_dwpw(B, H, W, C, Co, s, 0, 10, ctypes.byref(ms)))
    pair = ms.value
    os.environ["SSD_TS_DUMP"] = path
    check(L.ssd_bench_dwpw(B, H, W, C, Co, s, 1, 10, ctypes.byref(ms)))      # 1: dwpw_stream.hip
    os.environ.pop("SSD_TS_DUMP")
    stream = ms.value
    ph = np.fromfile(path, dtype=np.int64).reshape(-1, 8).astype(np.float64)
    ph = ph[ph[:, 7] > 0]
    per_it = ph[:, :7].sum(0) / ph[:, 7].sum()
    gb = (B * H * W * C + B * (H // s) * (W // s) * Co) * 4 / 1e9
    gf = B * (H // s) * (W // s) * (2.0 * C * Co + 18.0 * C) / 1e9
    line = "Conv2d_%d dw s%d %dx%dx%d -> pw %d: pair %.3f ms, stream %.3f ms (%.2f TB/s, %.1f TFLOP/s algorithmic)" % (
        i, s, H, W, C, Co, pair, stream, gb / stream, gf / stream)
    print(line, flush=True)
    print("      stream kernel, cycles per iteration (tile x 32-channel slice), mean over %d blocks: wait %.0f, barrier1 %.0f, depthwise %.0f, "
          "barrier2 %.0f, dma issue %.0f, mfma %.0f, epilogue %.0f; iterations per block %.0f"
          % ((len(ph),) + tuple(per_it) + (ph[:, 7].mean(),)), flush=True)
