#!/bin/bash
# PMC passes (separate, --kernel-trace only) over scripts/nonident_cost.py: HBM bytes and issue mix of the resized-frame kernels
# (first_conv_gen_kernel = K1d under front_fuse = 0, front_kernel<1> = the fused first-layers launch with the gather) next to the
# identity ones.   usage (on the GPU box): bash scripts/pmc_nonident.sh   -> gpurun_out/pmc_nonident/summary.txt
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_nonident; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/scripts/nonident_cost.py 2 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python3 - <<'P'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_nonident"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "first_conv" in k or "front_" in k or "dwpw_stream_kernel<1, 1>" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k in agg:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(out + "/summary.txt", "w") as fo:
    fo.write("# per-launch means over scripts/nonident_cost.py 2 (16-frame launches: one per backbone chain); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction)\n")
    for k in sorted(agg):
        m = {c: sum(v) / len(v) for c, v in agg[k].items()}
        fo.write("%s  launches=%d avg_us(under the counter pass)=%.1f\n" % (k, len(dur.get(k, [])), sum(dur[k]) / max(len(dur[k]), 1) if dur.get(k) else float("nan")))
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            fo.write("    hbm_bytes_per_launch=%.4g (fetch_kb=%.4g write_kb=%.4g)\n" % ((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024, m["FETCH_SIZE"], m["WRITE_SIZE"]))
        for c in sorted(m):
            fo.write("    %-22s %.6g\n" % (c, m[c]))
print(open(out + "/summary.txt").read())
P
