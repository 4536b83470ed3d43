#!/bin/bash
# Kernel timeline of one batch-1 forward under given library options.  usage (on the GPU box): scripts/tl.sh <name> [option=value ...]
# -> gpurun_out/r04/tl_<name>.txt
R=$GRAFT_REPO_ROOT; N=$1; shift
O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tl_$N
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$N -- python3 $R/scripts/b1_loop.py 12 f32 "$@" > $O/tl_$N.out 2>&1
python3 $R/scripts/b1_timeline.py /tmp/tl_$N > $O/tl_$N.txt 2>&1
cd $R
