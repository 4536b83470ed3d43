#!/bin/bash
# Same-box A/B of a library option (ssd_set_option; bench.py --option): bench.py with and without it, alternating, three rounds.
#   usage (on the GPU box): bash scripts/ab_opt.sh "key=value [key2=value2]" [bench.py arguments]
SW=""; for kv in $1; do SW="$SW --option $kv"; done; shift
ARGS="--no-other-precision --no-cpu-baseline --no-latency --no-shufflenet --no-traffic --sustained-seconds 0 --steps 20 $*"
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]
print(sys.argv[1].ljust(8), round(d["value"],1), round(d["ms_per_step"],3), {a:round(b,3) for a,b in k.items() if b})'
for i in 1 2 3; do
  python bench.py $ARGS 2>/dev/null | python -c "$P" default
  python bench.py $ARGS $SW 2>/dev/null | python -c "$P" option
done
