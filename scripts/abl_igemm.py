import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch, ssd_amd
from ssd_amd._lib import check
ssd_amd._lib.use_diag(); L = ssd_amd.lib()
names={0:"full kernel",10:"no global loads / LDS writes in the K loop",11:"+ no fragment reads",12:"+ no barrier (bare MFMAs + pro/epilogue)"}
for rnd in range(2):
  for (nm,H,W,Cin,Cout,k,pyr) in [("tower 3x3 256->256 5 levels",80,112,256,256,3,1),("pw 512->512 40x56",40,56,512,512,1,0)]:
    for t in (0,10,11,12):
        ms, gf = ctypes.c_double(), ctypes.c_double()
        check(L.ssd_bench_conv(32, H, W, Cin, Cout, k, 1, t, 10, pyr, ctypes.byref(ms), ctypes.byref(gf)))
        print("%-28s %-50s %8.3f ms %5.1f%%" % (nm, names[t], ms.value, gf.value/ms.value/157.3*100), flush=True)
