#!/bin/bash
# Which leg of bench.py's sequence moves the batch-1 latency measured behind it?  (one box, the legs switched off one at a time)
set -e
mkdir -p gpurun_out
run() { tag=$1; shift; python bench.py --no-shufflenet --no-cpu-baseline --no-traffic "$@" > gpurun_out/lat_bisect_$tag.json 2> gpurun_out/lat_bisect_$tag.err;
  python - "$tag" <<'P'
import json, sys
t = sys.argv[1]
d = json.loads(open("gpurun_out/lat_bisect_%s.json" % t).read().strip().splitlines()[-1])
l = d["latency_batch1"]; m = d.get("latency_mixed_sizes", {})
print("%-28s value %.1f  b1 p50 %.4f mean %.4f std %.4f  f16x3 p50 %.4f  mixed alone 640x896 %s  numa %s" % (t, d["value"], l["p50_ms"], l["mean_ms"], l["std_ms"],
      l["by_precision"]["f16x3"]["p50_ms"], m.get("alone_p50_ms", {}).get("640x896", {}).get("p50_ms"), d["numa_node_bound"]), flush=True)
P
}
run full
run no_sustained --sustained-seconds 0
run no_other --no-other-precision
run neither --sustained-seconds 0 --no-other-precision
run full_again
