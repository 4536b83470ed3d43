"""Does the CPU side of a batch-1 Detector call depend on which NUMA node the Python thread runs on?
usage: python scripts/numa_probe.py   (prints the GPU's node, then Detector p50 unbound / bound to each node)"""
import os, sys, time, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ssd_amd
P = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
     "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 640}


def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if "-" in part:
            a, b = part.split("-"); out.extend(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


pr = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
try:
    gnode = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
except OSError:
    gnode = -1
nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    nodes[int(d.rsplit("node", 1)[1])] = cpulist(open(d + "/cpulist").read())
allowed = sorted(os.sched_getaffinity(0))
print("GPU %s on NUMA node %d; nodes: %s; this process may run on %d CPUs (%d..%d)" %
      (bdf, gnode, {k: len(v) for k, v in nodes.items()}, len(allowed), allowed[0], allowed[-1]), flush=True)
img = np.random.default_rng(0).integers(0, 256, (640, 896, 3), dtype=np.uint8)


def run(tag):
    det = ssd_amd.Detector(ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5), config=P)
    ts = []
    for _ in range(160):
        t0 = time.perf_counter(); det(img, score_threshold=0.5); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-44s p50 %.4f ms  p10 %.4f" % (tag, np.percentile(ts[10:], 50), np.percentile(ts[10:], 10)), flush=True)
    det.engine.close()


run("unbound")
for n, cpus in nodes.items():
    c = [x for x in cpus if x in allowed]
    if not c:
        print("node %d: no allowed CPU" % n); continue
    os.sched_setaffinity(0, c)
    run("bound to node %d%s (%d CPUs)" % (n, " = the GPU's" if n == gnode else "", len(c)))
os.sched_setaffinity(0, allowed)
run("unbound again")
