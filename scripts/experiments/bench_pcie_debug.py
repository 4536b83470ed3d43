"""Debug aid: bench.py with Engine.detect_stream's host-side blocking points time-stamped for the first batches (stderr).
Round 6 finding: the first two batches of a batch shape spend ~0.15 s each creating their buffer set (pinning 32 MB of host memory),
during which finished results wait to be collected -- bench.py's pcie_inclusive leg now warms the stream up and times interior results.
    python scripts/experiments/bench_pcie_debug.py --no-shufflenet --no-cpu-baseline --no-latency --sustained-seconds 0 > /dev/null"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench, ssd_amd
from importlib import import_module
ssdmod = import_module("ssd_amd.ssd")


def detect_stream(self, batches):
    dev = torch.device("cuda", self.device)
    if self._copy_streams is None:
        self._copy_streams = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
    s_in, s_out = self._copy_streams
    s_c = torch.cuda.current_stream(dev)
    pending = None
    k = 0
    T0 = time.perf_counter()

    def log(what):
        if k <= 4:
            print("   k=%d %-28s %8.1f ms" % (k, what, (time.perf_counter() - T0) * 1e3), file=sys.stderr, flush=True)

    def collect(p):
        oslot, e = p
        e["d2h"].synchronize()
        log("collect: d2h event done")
        return tuple(np.array(v) for v in oslot["host"])

    for images in batches:
        images = np.asarray(images)
        key = tuple(images.shape)
        j = k & 1
        log("iteration start")
        islot, dev_in, pin_in = self._in_slot(key, index=1 + j, pinned=True)
        oslot = self._out_slot(key[0], index=1 + j)
        log("slots ready")
        e = islot["events"]
        if e is None:
            e = islot["events"] = {n: torch.cuda.Event() for n in ("h2d", "cmp", "d2h")}
            for n in e:
                e[n].record(s_c)
        e["h2d"].synchronize()
        log("h2d event waited")
        e["d2h"].synchronize()
        log("d2h event waited")
        np.copyto(pin_in.numpy(), images)
        log("staged")
        with torch.cuda.stream(s_in):
            s_in.wait_event(e["cmp"])
            dev_in.copy_(pin_in, non_blocking=True)
            e["h2d"].record(s_in)
        s_c.wait_event(e["h2d"])
        log("upload enqueued")
        self.forward(dev_in, records=oslot["block"])
        log("forward enqueued")
        e["cmp"].record(s_c)
        with torch.cuda.stream(s_out):
            s_out.wait_event(e["cmp"])
            oslot["pin_out"].copy_(oslot["block"], non_blocking=True)
            e["d2h"].record(s_out)
        log("d2h enqueued")
        if pending is not None:
            yield collect(pending)
        pending = (oslot, e)
        k += 1
    if pending is not None:
        yield collect(pending)


ssdmod.Engine.detect_stream = detect_stream
bench.main(sys.argv[1:])
