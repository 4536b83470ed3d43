"""Stand-in engine for tests/test_bench_launcher.py: runs bench.py's own launcher, sharding, step loop,
all-gather and JSON line on CPU (gloo) with an engine that fabricates detections from the frames'
bytes.  Test infrastructure only -- nothing in the product imports this."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class StubEngine:
    T = 2000

    def __init__(self, params, weights, device):
        self.calls = 0

    def forward(self, images):
        B = images.shape[0]
        # deterministic function of the shard's own pixels, so the gathered result can be checked per image
        key = images.reshape(B, -1)[:, :16].to(torch.int32).sum(dim=1)
        boxes = key.view(B, 1, 1).expand(B, self.T, 4).to(torch.float32).contiguous()
        labels = key.view(B, 1).expand(B, self.T).contiguous().to(torch.int32)
        scores = (key.view(B, 1).expand(B, self.T).to(torch.float32) / 4096.0).contiguous()
        num = (key % 7).to(torch.int32)
        self.calls += 1
        return boxes, labels, scores, num

    def profile_reset(self): pass
    def profile_enable(self, on=True): pass
    def profile_read(self): return {}
    def status(self): return 0
    def set_precision(self, p): pass
    def close(self): pass


if __name__ == "__main__":
    bench.H, bench.W = 8, 8          # tiny frames: this run checks plumbing, not throughput
    bench.main(engine_factory=StubEngine, backend="gloo", script=os.path.abspath(__file__))
