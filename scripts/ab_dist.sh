#!/bin/bash
# Same-box A/B: bench.py plain against bench.py --force-dist (the whole collective path on RCCL with one rank), alternating
# runs.  The throughput comparison that used to be asserted inside tests/test_gpu_serving.py lives here: it is a
# measurement, not a parity property.   usage: scripts/ab_dist.sh [rounds]
R=${1:-3}
C="--steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-shufflenet --no-other-precision --no-traffic"
for i in $(seq $R); do
  python bench.py $C | python -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith("{")][0]); print("plain      %.1f img/s" % d["value"])'
  python bench.py $C --force-dist | python -c 'import json,sys; d=json.loads([l for l in sys.stdin if l.startswith("{")][0]); print("force-dist %.1f img/s  ranks %s" % (d["value"], d["ranks_seen"]))'
done
