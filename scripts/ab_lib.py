"""Same-box A/B of two builds of the library: run bench.py's main() against an explicitly named .so (built from another
revision of csrc/, kept under build_tmp/ or gpurun_out/) instead of csrc/libssd_hip.so.
usage: python scripts/ab_lib.py <path/to/libssd_hip.so> [bench.py arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssd_amd
from ssd_amd import _lib
path = os.path.abspath(sys.argv[1])
assert os.path.exists(path), path
_lib.build = lambda *a, **k: path          # no freshness check: the named file is what gets loaded
import ctypes
_other = ctypes.CDLL(path)                 # an older build may lack entry points the tree has since added: bind what it has
for name in [n for n in _lib.SIGNATURES if not hasattr(_other, n)]:
    del _lib.SIGNATURES[name]
import bench
sys.exit(bench.main(sys.argv[2:]))
