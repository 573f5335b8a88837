#!/bin/bash
# HBM-side traffic counters of the render kernel for any bench configuration (separate --pmc passes, kernel trace only):
#   bash tools/pmc_mem.sh TAG [bench.py flags]      e.g.  bash tools/pmc_mem.sh v4 --views 4
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcmem_$1
mkdir -p $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc ${@:2} > $OUT/pass$i.log 2>&1
  tail -1 $OUT/pass$i.log | cut -c1-200
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob('$OUT/pass*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if 'render_pass' in row['Kernel_Name']:
                agg[(row['Kernel_Name'][:64], row['Counter_Name'])][0] += float(row['Counter_Value'])
                agg[(row['Kernel_Name'][:64], row['Counter_Name'])][1] += 1
        for (kn, cn), (v, n) in sorted(agg.items()):
            print(f'$1 {kn:64s} {cn:16s} per-launch {v/n:.4e}  (n={n})')
PY
