#!/bin/bash
# per-layer kernel durations of one CostRegNet forward: bash tools/trace_costreg.sh TAG h w
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_costreg_$1
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/time_cost_volume.py $2 $3 costregnet > $OUT.log 2>&1
tail -3 $OUT.log
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in ('conv3d', 'avgpool', 'upsample', 'cout1'))]
last = sel[-15:]
tot = 0
for r in last:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print(f"{r['Kernel_Name'][:70]:70s} grid={r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']:>3s} {d:9.1f} us")
print('sum', tot, 'us')
PY
