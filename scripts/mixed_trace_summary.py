"""HIP API call counts of two `rocprofv3 --hip-trace --stats` runs of scripts/mixed_trace.py (A: few cycles, B: many): per API
the calls in A, in B, and the difference per steady-state Detector call.

    python scripts/mixed_trace_summary.py A_hip_api_stats.csv B_hip_api_stats.csv calls_A calls_B
"""
import csv
import sys


def load(path):
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            out[row["Name"]] = int(row["Calls"])
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
ca, cb = int(sys.argv[3]), int(sys.argv[4])
print("HIP API calls of two traced runs of scripts/mixed_trace.py: %d and %d steady-state Detector calls over 13 image sizes" % (ca, cb))
print("(after one warm-up pass that builds the 7 layer plans); per-call = (B - A) / %d\n" % (cb - ca))
print("%-40s %10s %10s %12s" % ("HIP API", "run A", "run B", "per call"))
for name in sorted(set(a) | set(b), key=lambda n: -(b.get(n, 0) - a.get(n, 0))):
    d = b.get(name, 0) - a.get(name, 0)
    print("%-40s %10d %10d %12.3f" % (name, a.get(name, 0), b.get(name, 0), d / (cb - ca)))
zero = [n for n in sorted(set(a) | set(b)) if a.get(n, 0) == b.get(n, 0)]
print("\nnot called in steady state (same count in both runs): " + ", ".join(zero))
for must in ("hipDeviceSynchronize", "hipMalloc", "hipFree", "hipHostMalloc", "hipHostFree", "hipMemset", "hipMemcpy", "hipEventCreateWithFlags",
             "hipStreamCreateWithFlags"):
    print("%-28s steady-state calls: %d" % (must, b.get(must, 0) - a.get(must, 0)))
