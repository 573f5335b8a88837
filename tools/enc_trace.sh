#!/bin/bash
# kernel-trace of the per-frame 2-D encoders (torch/MIOpen): bash tools/enc_trace.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/enct
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/time_encoders.py > $OUT.log 2>&1
cat $OUT.log | tail -6
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$f')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time ms', tot/1e6, 'kernels', len(rows))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:25]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} share={100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
rm -rf $OUT
