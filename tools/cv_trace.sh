#!/bin/bash
# kernel-trace timing of the cost-volume kernels: bash tools/cv_trace.sh [h w]
cd /tmp && export TMPDIR=/tmp
H=${1:-128}; W=${2:-256}
OUT=$GRAFT_REPO_ROOT/gpurun_out/cvt_${H}x${W}
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/time_cost_volume.py $H $W costregnet > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
grep -E "cost_volume|hypoth" $f | cut -c1-170
