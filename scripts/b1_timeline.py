"""Kernel timeline of ONE batch-1 forward out of a `rocprofv3 --kernel-trace --output-format csv` directory.
usage: python scripts/b1_timeline.py <trace dir> [which forward from the end, default 2]
A forward = the kernels from a first_conv* / front_kernel launch up to and including the next post_pack_kernel.  Prints
start (us from the forward's first kernel), duration, kernel, grid, queue; then the span and per-kernel-family sums."""
import collections, csv, glob, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]),
                     int(r["Workgroup_Size_X"]), int(r["Queue_Id"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"])))
rows.sort()
ends = [i for i, r in enumerate(rows) if "post_pack" in r[2]]
if len(ends) < back + 1:
    sys.exit("not enough forwards in the trace")
hi = ends[-back]
lo = ends[-back - 1] + 1
while lo < hi and "first_conv" not in rows[lo][2] and "front_kernel" not in rows[lo][2] and "front_pool" not in rows[lo][2]:
    lo += 1
fw = rows[lo:hi + 1]
t0 = fw[0][0]
print("# start us, duration us, kernel, blocks x threads, queue, VGPRs, LDS bytes")
fam = collections.defaultdict(float)
for s, e, k, g, wg, q, vg, lds in fw:
    print("%8.1f %8.1f  %-46s %6d x %-4d q%d  v%-3d lds %d" % ((s - t0) / 1e3, (e - s) / 1e3, k[:46], g // max(wg, 1), wg, q, vg, lds))
    fam[k.split("<")[0]] += (e - s) / 1e3
print("# forward span %.1f us (first kernel start -> last kernel end), %d kernels" % ((fw[-1][1] - t0) / 1e3, len(fw)))
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    print("#   sum of durations %-28s %8.1f us" % (k, v))
