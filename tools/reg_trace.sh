#!/bin/bash
# kernel-trace of one regulariser timing run: bash tools/reg_trace.sh unet2|costregnet [h w]
cd /tmp && export TMPDIR=/tmp
NET=${1:-unet2}; H=${2:-128}; W=${3:-256}
OUT=$GRAFT_REPO_ROOT/gpurun_out/regt_${NET}_${H}x${W}
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/time_cost_volume.py $H $W $NET > $OUT.log 2>&1
grep -E "UNet2|CostRegNet" $OUT.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open('$f')) if 'pgrf' in r['Name']]
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:12]:
    print(f"{r['Name'][:78]:78s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} share={100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
cp $f $GRAFT_REPO_ROOT/gpurun_out/regt_${NET}_${H}x${W}_kernel_stats.csv
rm -rf $OUT
