#!/bin/bash
# Same-box A/B of the batch-1 latency (bench.py's latency_batch1 through Detector.__call__) of the library in the tree against another
# build of it (scripts/ab_lib.py), alternating, three rounds.   usage (on the GPU box): bash scripts/ab_b1_lib.sh <path/to/other/libssd_hip.so>
OTHER=${1:-scripts/experiments/bin/libssd_hip_prev.so}
ARGS="--no-other-precision --no-cpu-baseline --no-shufflenet --no-traffic --sustained-seconds 0 --steps 5 --warmup 2"
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d["latency_batch1"]["by_precision"]["f32"]
print(sys.argv[1].ljust(6), "p50 %.4f mean %.4f ms" % (l["p50_ms"], l["mean_ms"]), d["latency_batch1"].get("segments_p50_us"))'
for i in 1 2 3; do
  python bench.py $ARGS 2>/dev/null | python -c "$P" tree
  python scripts/ab_lib.py $OTHER $ARGS 2>/dev/null | python -c "$P" other
done
