#!/bin/bash
# Kernel timeline of one full-size forward.  usage (on the GPU box): scripts/tl_net.sh <name> <mobilenet|shufflenet> [batch] [option=value ...]
# -> gpurun_out/r05/tl_<name>.txt
R=$GRAFT_REPO_ROOT; N=$1; shift
O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tl_$N
NET=$1; shift; B=$1; shift
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$N -- python3 $R/scripts/net_loop.py $NET $B 6 "$@" > $O/tl_$N.out 2>&1
python3 $R/scripts/b1_timeline.py /tmp/tl_$N > $O/tl_$N.txt 2>&1
cd $R
