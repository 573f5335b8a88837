#!/bin/bash
# kernel statistics of the training bench (rocprofv3 --kernel-trace --stats), the top N kernels:   bash tools/train_stats.sh [N]
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the gpurun box exports it) or run from the repo root with GRAFT_REPO_ROOT=$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trs
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trs -- python3 $R/bench.py --train --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
f=$(find $R/gpurun_out/trs -name "*kernel_stats.csv" | head -1)
python3 - "$f" "${1:-24}" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2])]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(8), ("%.2f" % (float(r["TotalDurationNs"]) / 1e6)).rjust(8), r["Percentage"])
PY
rm -rf $R/gpurun_out/trs
