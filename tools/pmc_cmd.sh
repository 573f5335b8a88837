#!/bin/bash
# PMC passes (counters + kernel trace only) for any python script:  bash tools/pmc_cmd.sh TAG KERNEL_SUBSTR script.py [args...]
# Prints per-launch averages of every counter for kernels whose name contains KERNEL_SUBSTR.
cd /tmp && export TMPDIR=/tmp
TAG=$1; KSUB=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcc_$TAG
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_COUNT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 "$@" > $OUT/pass$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob('$OUT/pass*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if '$KSUB' in row['Kernel_Name']:
                k = (row['Kernel_Name'][:70], row['Counter_Name'])
                agg[k][0] += float(row['Counter_Value']); agg[k][1] += 1
        for (kn, cn), (v, n) in sorted(agg.items()):
            print(f'{kn:70s} {cn:30s} per-launch {v/n:.4e}  (n={n})')
PY
