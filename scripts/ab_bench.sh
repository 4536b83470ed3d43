#!/bin/bash
# Same-box A/B of the library in the tree against another build of it (scripts/ab_lib.py), alternating, three rounds.
#   usage (on the GPU box): bash scripts/ab_bench.sh <path/to/other/libssd_hip.so> [bench.py arguments]
OTHER=${1:-scripts/experiments/bin/libssd_hip_prev.so}; shift
ARGS="--no-other-precision --no-cpu-baseline --no-latency --no-shufflenet --no-traffic --sustained-seconds 0 --steps 20 $*"
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]
print(sys.argv[1].ljust(6), round(d["value"],1), round(d["ms_per_step"],3), {a:round(b,3) for a,b in k.items() if b})'
for i in 1 2 3; do
  python bench.py $ARGS 2>/dev/null | python -c "$P" tree
  python scripts/ab_lib.py $OTHER $ARGS 2>/dev/null | python -c "$P" other
done
