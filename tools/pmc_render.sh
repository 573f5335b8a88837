#!/bin/bash
# PMC passes for the render kernel (run on the GPU box through gpurun). Each pass is its own rocprofv3 run
# with --kernel-trace only (no sys/hip traces alongside counters).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc ${@:2} > $OUT/pass$i.log 2>&1
  tail -2 $OUT/pass$i.log | cut -c1-300
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob('$OUT/pass*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if 'render_pass' in row['Kernel_Name'] or 'conv3d' in row['Kernel_Name'] or 'cost_volume' in row['Kernel_Name']:
                k = (row['Kernel_Name'][:60], row['Counter_Name'])
                agg[k][0] += float(row['Counter_Value']); agg[k][1] += 1
        for (kn, cn), (v, n) in sorted(agg.items()):
            print(f'{kn:60s} {cn:28s} per-launch {v/n:.4e}  (n={n})')
PY
