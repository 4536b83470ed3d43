import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["traffic"], r.get("traffic_committed"), r["traffic_source"][:60], r.get("traffic_live_error"))
print(d["latency_batch1"]["p50_ms"], d["latency_batch1"]["runs_p50_ms"], d["latency_mixed_sizes"]["mean_ms"], d["throughput_mixed_sizes"]["detect_many_img_s"], d["shufflenet_config4"]["value"], d["cpu_baseline"]["value"])
