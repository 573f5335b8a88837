#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/encht
rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/scratch/enc_hip_only.py > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open('$f'))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last forward = last len/6 launches
n=len(rows)//6
last=rows[-n:]
t0=int(last[0]['Start_Timestamp']); t1=int(last[-1]['End_Timestamp'])
print('launches per forward', n, 'span ms', (t1-t0)/1e6, 'sum kernel ms', sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in last)/1e6)
for r in last:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print(f"{d:8.1f} us  grid={r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']:<3s} {r['Kernel_Name'][:90]}")
PY
rm -rf $OUT
