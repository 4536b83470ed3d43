"""Shader clock and socket power while the detector runs (sysfs samples every 20 ms beside a loop of forwards).
usage: python scripts/clock_probe.py [seconds per phase]
Phases: idle, batch-1 synchronised forwards, batch-32 back-to-back forwards.  Tells a clock-bound kernel (MFMA pipe full,
chip below its nominal 2.4 GHz) from a stalled one."""
import glob, os, re, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, ssd_amd
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
P = {"backbone": "mobilenet", "depth_multiplier": 1.0, "num_classes": 80, "score_threshold": 0.15,
     "iou_threshold": 0.6, "max_boxes_per_class": 25, "min_dimension": 640}


def find(pattern):
    for p in sorted(glob.glob(pattern)):
        try:
            open(p).read()
            return p
        except OSError:
            pass
    return None


pr = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
dev = "/sys/bus/pci/devices/" + bdf
print("device", pr.name, bdf, "exists" if os.path.isdir(dev) else "NOT in sysfs", flush=True)
if not os.path.isdir(dev):
    dev = "/sys/class/drm/card*/device"
sclk = find(dev + "/pp_dpm_sclk")
power = find(dev + "/hwmon/hwmon*/power1_average") or find(dev + "/hwmon/hwmon*/power1_input")
print("sclk file:", sclk, " power file:", power, flush=True)
if sclk:
    print("pp_dpm_sclk now:\n" + open(sclk).read(), flush=True)


def sample():
    mhz = w = None
    if sclk:
        m = re.search(r"(\d+)Mhz \*", open(sclk).read())
        mhz = int(m.group(1)) if m else None
    if power:
        try:
            w = int(open(power).read()) / 1e6
        except (OSError, ValueError):
            w = None
    return mhz, w


stop, samples = False, []


def sampler():
    while not stop:
        samples.append(sample())
        time.sleep(0.02)


def phase(name, fn):
    global stop, samples
    stop, samples = False, []
    t = threading.Thread(target=sampler)
    t.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < SECS:
        fn()
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop = True
    t.join()
    mh = sorted(s[0] for s in samples if s[0] is not None)
    pw = sorted(s[1] for s in samples if s[1] is not None)
    med = lambda v: v[len(v) // 2] if v else None
    print("%-22s %6d calls  %.3f ms/call   sclk MHz min/median/max %s/%s/%s   power W median %s max %s   (%d samples)" %
          (name, n, dt / max(n, 1) * 1e3, mh[0] if mh else None, med(mh), mh[-1] if mh else None, med(pw), pw[-1] if pw else None, len(samples)), flush=True)


e = ssd_amd.Engine(P, ssd_amd.synthetic_weights(P, seed=0, logits_bias=-7.5))
img1 = torch.randint(0, 256, (1, 640, 896, 3), dtype=torch.uint8).cuda()
img32 = torch.randint(0, 256, (32, 640, 896, 3), dtype=torch.uint8).cuda()
for _ in range(3):
    e.forward(img1); e.forward(img32)
torch.cuda.synchronize()
phase("idle", lambda: time.sleep(0.05))


def b1():
    e.forward(img1)
    torch.cuda.synchronize()


def b32():
    e.forward(img32)
    torch.cuda.synchronize()


phase("batch 1 forwards", b1)
phase("batch 32 forwards", b32)
phase("batch 1 forwards", b1)
