"""Prints the kernel sequence of ONE forward from a rocprofv3 kernel-trace csv (last forward)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "first_conv" in r["Kernel_Name"]]
lo = starts[-2] if len(starts) > 1 else starts[-1]
hi = starts[-1] if len(starts) > 1 else len(rows)
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    print("%9.3f %9.3f  %-40s grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, n[:40], r["Grid_Size_X"]))
