#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_unet_$1
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/time_cost_volume.py $2 $3 > $OUT.log 2>&1
tail -4 $OUT.log
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last UNet2 forward = the last 27 dispatches of conv/pool/up kernels
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in ('conv3d', 'avgpool', 'upsample'))]
last = sel[-27:]
tot = 0
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print(f"{r['Kernel_Name'][:60]:60s} grid={r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']:>3s} {d:9.1f} us")
print('sum', tot, 'us')
PY
