import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, ssd_amd
from ssd_amd._lib import check
ssd_amd._lib.use_diag()
L = ssd_amd.lib()
B = 32
def run(name, Cout, tile):
    ms, gf = ctypes.c_double(), ctypes.c_double()
    check(L.ssd_bench_conv(B, 80, 112, 256, Cout, 3, 1, tile, 10, 1, ctypes.byref(ms), ctypes.byref(gf)))
    print("%-30s %8.3f ms %7.1f TFLOP/s" % (name, ms.value, gf.value / ms.value), flush=True)
    return ms.value
for r in range(2):
    a = run("logits 480 tile 128x96", 480, 6)
    b = run("logits 384 tile 128x128", 384, 0)
    c = run("logits 96 tile 128x96", 96, 6)
    d = run("logits 512(480 padded) 128x128", 480, 0)
    e = run("logits 256 tile 128x128", 256, 0)
    f = run("logits 224 tile 128x96?", 224, 0)
    print("split 384+96: %.3f ms vs %.3f" % (b + c, a))
