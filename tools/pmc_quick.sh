#!/bin/bash
# one SQ counter pass on bench.py for a library variant: tools/pmc_quick.sh TAG [variant]
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$1
if [ -n "$2" ] && [ "$2" != base ]; then export PANOGRF_HIP_LIB=$GRAFT_REPO_ROOT/panogrf_amd/libpanogrf_hip_$2.so PGRF_ALLOW_ABLATION_LIB=1; fi
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc > $OUT.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d ${OUT}b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc > ${OUT}b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ['$OUT', '${OUT}b']:
  for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if 'render_pass' in row['Kernel_Name']:
            agg[row['Counter_Name']][0] += float(row['Counter_Value']); agg[row['Counter_Name']][1] += 1
    tiles = 1048576.0
    for cn, (v, n) in sorted(agg.items()):
        print(f'$1 $2 {cn:28s} per-launch {v/n:.4e}   per-tile x4: {4*v/n/tiles:10.0f}')
PY
