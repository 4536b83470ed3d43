"""Condenses a scripts/collect_profiles.sh output directory into the files committed under
profiles/: kernel_stats.csv (rocprofv3 --stats), pmc_summary.txt (per-kernel means of the
PMC passes, with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md applied)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out = sys.argv[1]
stats = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], out + "/kernel_stats.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:56]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:56]
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
# union of the traced [start, end] intervals per kernel: the two head towers run concurrently on
# two streams, so a kernel's own duration includes time shared with its twin
union = {}
iv = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        iv[r["Kernel_Name"].split("(")[0][:56]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in iv.items():
    v.sort()
    tot, lo, hi = 0, None, None
    for a_, b_ in v:
        if hi is None:
            lo, hi = a_, b_
        elif a_ <= hi:
            hi = max(hi, b_)
        else:
            tot += hi - lo
            lo, hi = a_, b_
    if hi is not None:
        tot += hi - lo
    union[k] = tot / 1e6 / len(v)
def union_ms(intervals):
    intervals = sorted(intervals)
    tot, lo, hi = 0, None, None
    for a_, b_ in intervals:
        if hi is None:
            lo, hi = a_, b_
        elif a_ <= hi:
            hi = max(hi, b_)
        else:
            tot += hi - lo
            lo, hi = a_, b_
    if hi is not None:
        tot += hi - lo
    return tot / 1e6


# kernel_stats.csv again with one more column: the union of each kernel's launch intervals per launch (kernels of the
# two head towers overlap on two streams: their own average duration counts the shared GPU twice)
if stats:
    full = collections.defaultdict(list)
    for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            full[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
    with open(out + "/kernel_stats_union.csv", "w", newline="") as fo:
        wr = csv.writer(fo)
        wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "UnionNsPerLaunch"])
        for r in rows:
            v = full.get(r["Name"], [])
            wr.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                         "%.1f" % (union_ms(v) * 1e6 / max(len(v), 1)) if v else ""])

nfwd = max(1, sum(len(v) for k, v in iv.items() if "post_pack" in k))      # one per forward (the backbone may run as two chains)
import re
# every instance whose TAPS template argument is 9: igemm_kernel<WM, WN, wm, wn, 9, ...> (all tile forms, the deep-prefetch
# 64x64 instance included), igemm_lat_kernel<PT, CT, 9, ...> (fpn p6 / p7 at batch 1-2), igemm16_kernel<9, ...>
is3x3 = lambda k: bool(re.search(r"igemm_kernel<\d+, \d+, \d+, \d+, 9,", k) or re.search(r"igemm_lat_kernel<\d+, \d+, 9,", k) or "igemm16_kernel<9" in k)
c3 = [x for k, v in iv.items() if is3x3(k) for x in v]
class_line = ("all 3x3 igemm kernels (bench.py class conv3x3_mfma): %d launches in %d forwards, union %.3f ms per forward"
              " = %.4f ms per launch\n" % (len(c3), nfwd, union_ms(c3) / nfwd, union_ms(c3) / max(len(c3), 1)))
with open(out + "/pmc_summary.txt", "w") as fo:
    fo.write("# " + class_line)
    fo.write("# per-dispatch means; FETCH/WRITE in KB as reported; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024\n")
    fo.write("# (gfx950: FETCH_SIZE reports half of a wide coalesced read stream, MI355X_MICROARCH.md section HBM)\n")
    for k in sorted(agg, key=lambda k: -sum(dur.get(k, [0]))):
        d = agg[k]
        m = {c: sum(v) / len(v) for c, v in d.items()}
        line = "%s  launches(traced)=%d avg_ms=%.4f union_ms_per_launch=%.4f\n" % (
            k, len(dur.get(k, [])), (sum(dur[k]) / len(dur[k])) if dur.get(k) else float("nan"), union.get(k, float("nan")))
        fo.write(line)
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            fo.write("    hbm_bytes_per_launch=%.4g (fetch_kb=%.4g write_kb=%.4g)\n" % ((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024, m["FETCH_SIZE"], m["WRITE_SIZE"]))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
            fo.write("    mfma_util=%.3f (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs))\n" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)))
        for c in sorted(m):
            fo.write("    %-28s %.6g\n" % (c, m[c]))
# traffic of the dominant kernel for bench.py's roofline.traffic: the 256x256-tile f16x3 kernel when the run
# used it (bench.py's default mode), else the exact-fp32 3x3 implicit GEMM; merged into one file by mode
dom16 = [k for k in agg if "igemm16_kernel<9" in k]
dom32 = [k for k in agg if "igemm_kernel<2, 2, 2, 2, 9, 0, 0" in k or k.rstrip().endswith("igemm_kernel<2, 2, 2, 2, 9, 0>")]
mode, dom = ("f16x3", dom16) if dom16 else ("f32", dom32)
if dom and "FETCH_SIZE" in agg[dom[0]] and "WRITE_SIZE" in agg[dom[0]]:
    f = agg[dom[0]]["FETCH_SIZE"]; w = agg[dom[0]]["WRITE_SIZE"]
    t = {"kernel": dom[0], "hbm_bytes_per_launch": (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024,
         "launches_averaged": len(f), "correction": "FETCH_SIZE x2 (gfx950), KB -> bytes",
         "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py --precision " + mode}
    json.dump({mode: t}, open(out + "/traffic.json", "w"), indent=1)
    print(t)
print(open(out + "/pmc_summary.txt").read()[:3000])
if os.path.exists(out + "/bench.json"):
    print(open(out + "/bench.json").read())
