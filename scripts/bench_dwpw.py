"""Depthwise+pointwise blocks of MobileNet-v1 at 640x896: fused kernel vs the two-kernel pair,
and the phase timestamps of the fused kernel.  usage: python scripts/bench_dwpw.py [B] [layers]"""
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
          5: (80, 112, 256, 256, 1), 6: (80, 112, 256, 512, 2)}
path = "/tmp/ts_dwpw.bin"
for i in sel:
    H, W, C, Co, s = LAYERS[i]
    ms = ctypes.c_double()
    check(L.ssd_bench_dwpw(B, H, W, C, Co, s, 0, 10, ctypes.byref(ms)))
    pair = ms.value
    os.environ["SSD_TS_DUMP"] = path
    check(L.ssd_bench_dwpw(B, H, W, C, Co, s, 1, 10, ctypes.byref(ms)))
    os.environ.pop("SSD_TS_DUMP")
    t = np.fromfile(path, dtype=np.int64).reshape(-1, 5).astype(np.float64) * 0.01
    d = np.diff(t, axis=1).mean(0)
    gb = (B * H * W * C + B * (H // s) * (W // s) * Co) * 4 / 1e9
    print("Conv2d_%d dw s%d %dx%dx%d -> pw %d: pair %.3f ms, fused %.3f ms (%.2f TB/s algorithmic); %d blocks, phases us: "
          "depthwise->LDS %.2f, first B stage %.2f, K loop %.2f, epilogue %.2f, life %.2f"
          % (i, s, H, W, C, Co, pair, ms.value, gb / ms.value, len(t), d[0], d[1], d[2], d[3], (t[:, 4] - t[:, 0]).mean()), flush=True)
