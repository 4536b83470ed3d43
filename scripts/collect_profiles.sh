#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the bench line of the command WITHOUT a profiler (bench.json -- the clean
# measurement), the rocprofv3 kernel-trace summary of the same command (bench_traced.json is the line printed UNDER the
# profiler: perturbed, for cross-checking kernel names only) and separate PMC passes (HBM traffic, MFMA busy) for it.
#   usage: BENCH_ARGS="..." bash scripts/collect_profiles.sh <tag>      outputs under gpurun_out/<tag>/
R=$GRAFT_REPO_ROOT; TAG=${1:-r01}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
# BENCH_ARGS selects what the traced / counted runs measure, e.g. "--no-other-precision --no-shufflenet" (mode f32 of the
# headline workload alone) or "--config shufflenet --no-other-precision" (BASELINE config 4 alone)
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-traffic --sustained-seconds 0 $BENCH_ARGS > $OUT/bench.json 2> $OUT/bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-traffic --sustained-seconds 0 $BENCH_ARGS > $OUT/bench_traced.json 2> $OUT/trace.err
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-traffic --sustained-seconds 0 $BENCH_ARGS > $OUT/pmc$i.json 2> $OUT/pmc$i.err
  echo "pmc pass $i rc=$?"
done
echo "collected $TAG"
cd $R
python3 scripts/summarize_profiles.py $OUT
