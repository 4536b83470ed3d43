#!/bin/bash
# usage: scripts/pmc.sh <outdir-under-gpurun_out> <python script + args...>
# Runs separate rocprofv3 --pmc passes (counter groups that fit the per-pass slots).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 "$@" > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write("   %-32s n=%d mean=%.6g\n" % (c, len(v), sum(v) / len(v)))
print(open(out + "/summary.txt").read())
PY
