#!/bin/bash
# Timeline of the LAST training step of the bench (every launch in start order, with its stream-queue and the gap to the previous
# launch's end on the same queue):   bash tools/train_timeline.sh
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the gpurun box exports it) or run from the repo root with GRAFT_REPO_ROOT=$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trl
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trl -- python3 $R/bench.py --train --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
f=$(find $R/gpurun_out/trl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<PY
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step = the launches after the last-but-one Adam (multi_tensor_apply) burst
idx = [i for i, r in enumerate(rows) if 'multi_tensor_apply' in r['Kernel_Name']]
# group bursts
bursts = []
for i in idx:
    if not bursts or i - bursts[-1][-1] > 40: bursts.append([i])
    else: bursts[-1].append(i)
start = bursts[-2][-1] + 1 if len(bursts) >= 2 else 0
end = bursts[-1][-1] + 1
step = rows[start:end]
t0 = int(step[0]["Start_Timestamp"])
print('launches in the step: %d, span %.3f ms' % (len(step), (int(step[-1]["End_Timestamp"]) - t0) / 1e6))
last_end = {}
busy = collections.Counter()
for r in step:
    q = r.get("Queue_Id", "?")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    busy[q] += e - s
    print('%8.1f us  q%-3s +%6.1f  %7.1f us  %s' % ((s - t0) / 1e3, q, gap, (e - s) / 1e3, r["Kernel_Name"][:70]))
print({q: round(v / 1e6, 3) for q, v in busy.items()})
PY
rm -rf $R/gpurun_out/trl
