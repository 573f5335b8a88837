#!/bin/bash
# per-launch durations of one kernel family in the LAST training step of the bench (launch order):   bash tools/train_trace.sh chain_fwd [count]
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the gpurun box exports it) or run from the repo root with GRAFT_REPO_ROOT=$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trt
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trt -- python3 $R/bench.py --train --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
f=$(find $R/gpurun_out/trt -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$1" "${2:-26}" <<PY
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3])
for r in rows[-n:]:
    print(r["Kernel_Name"][:50].ljust(50), "grid", r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), "lds", r.get("LDS_Block_Size"), "%.1f us" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf $R/gpurun_out/trt
