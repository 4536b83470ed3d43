"""Runs GPU tests on csrc/libssd_hip_chk.so (scripts/host_checked.sh: host code of the library under UBSan + libstdc++ assertions)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssd_amd._lib as L   # noqa: E402

path = os.path.join(os.path.dirname(L.lib_path()), "libssd_hip_chk.so")
if not os.path.exists(path):
    sys.exit("build it first: scripts/host_checked.sh build")
L._LIB_PATH = path
L.build = lambda *a, **k: path          # never rebuild over it
import pytest   # noqa: E402

args = sys.argv[1:] or ["tests/test_gpu_forward.py", "tests/test_gpu_shufflenet_b64.py", "tests/test_gpu_z_serving.py",
                        "tests/test_gpu_stages.py", "tests/test_gpu_eval_harness.py", "tests/test_gpu_f16x3.py"]
rc = pytest.main(args + ["-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"])
maps = open("/proc/self/maps").read()
print("loaded:", sorted({os.path.basename(ln.split()[-1]) for ln in maps.splitlines() if "libssd_hip" in ln or "ubsan" in ln}))
sys.exit(rc)
