#!/bin/bash
# Run ON THE GPU BOX (through gpurun): batch-1 forwards (scripts/b1_loop.py) under rocprofv3 -- the kernel timeline of the
# plan as it runs (three streams), the timeline with every kernel alone on one stream, and separate PMC passes of the
# one-stream loop (MFMA busy, LDS conflicts, traffic), condensed by summarize_profiles.py.
#   usage: bash scripts/collect_b1.sh <tag>      outputs under gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT; TAG=${1:-r04b1}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/trace3 $OUT/trace $OUT/pmc*
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace3 -- python3 $R/scripts/b1_loop.py 12 f32 > $OUT/loop3.out 2>&1
python3 $R/scripts/b1_timeline.py $OUT/trace3 > $OUT/timeline.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/b1_loop.py 12 f32 streams=1 > $OUT/loop1.out 2>&1
python3 $R/scripts/b1_timeline.py $OUT/trace > $OUT/timeline_single_stream.txt 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $R/scripts/b1_loop.py 12 f32 streams=1 > $OUT/pmc$i.out 2>&1
  echo "pmc pass $i rc=$?"
done
cd $R
python3 scripts/summarize_profiles.py $OUT > $OUT/summary.out 2>&1
echo "collected $TAG"
