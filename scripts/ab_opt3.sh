#!/bin/bash
# Same-box A/B of several values of ONE library option: bench.py default, then each value, alternating, two rounds.
#   usage (on the GPU box): bash scripts/ab_opt3.sh key "v1 v2 ..." [bench.py arguments]
K=$1; VALS=$2; shift; shift
ARGS="--no-other-precision --no-cpu-baseline --no-latency --no-shufflenet --sustained-seconds 0 --steps 20 $*"
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]
print(sys.argv[1].ljust(12), round(d["value"],1), round(d["ms_per_step"],3), {a:round(b,3) for a,b in k.items() if b})'
for i in 1 2; do
  python bench.py $ARGS 2>/dev/null | python -c "$P" default
  for v in $VALS; do python bench.py $ARGS --option $K=$v 2>/dev/null | python -c "$P" "$K=$v"; done
done
