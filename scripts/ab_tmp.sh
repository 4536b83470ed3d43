P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]
print(sys.argv[1].ljust(6), round(d["value"],1), round(d["ms_per_step"],3), {a:round(b,3) for a,b in k.items() if b})'
for i in 1 2 3; do
  python bench.py --no-other-precision --no-cpu-baseline --no-latency --no-shufflenet --steps 20 2>/dev/null | python -c "$P" new
  python scripts/ab_lib.py scripts/experiments/bin/libssd_hip_prev.so --no-other-precision --no-cpu-baseline --no-latency --no-shufflenet --steps 20 2>/dev/null | python -c "$P" prev
done
