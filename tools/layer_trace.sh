#!/bin/bash
# per-dispatch durations of the LAST regulariser forward of a timing run:  bash tools/layer_trace.sh [unet2|costregnet] [h w]
cd /tmp && export TMPDIR=/tmp
NET=${1:-unet2}; H=${2:-128}; W=${3:-256}
OUT=$GRAFT_REPO_ROOT/gpurun_out/layer_trace
rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/time_cost_volume.py $H $W $NET > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open('$f'))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
# last forward = from the last cost-volume-free stretch: take the trailing dispatches after the last 'cout1_gather' but one
idx=[i for i,n in enumerate(names) if 'cout1_gather' in n or 'conv3d_cout1' in n]
end=len(rows)
# find start: previous gather index + 1
gi=[i for i,n in enumerate(names) if 'cout1_gather' in n]
start=gi[-2]+1 if len(gi)>1 else 0
tot=0
for r in rows[start:gi[-1]+1]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    tot+=d
    n=r['Kernel_Name'].replace('void pgrf::','').replace('pgrf::','')[:60]
    print(f"{n:60s} grid={r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size','?'):>9s},{r.get('Grid_Size_Y','')} dur_us={d:9.1f}")
print('sum_us', tot)
PY
rm -rf $OUT
