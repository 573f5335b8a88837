#!/bin/bash
# SQ counters of the conv3d kernels during one UNet2 timing run: bash tools/pmc_unet.sh TAG
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcu_$1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/time_cost_volume.py 128 256 unet2 > $OUT.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:60] + ' grid=' + row['Grid_Size']
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); 
        if row['Counter_Name'] == 'SQ_INSTS_MFMA': cnt[k] += 1
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))[:8]:
        n = max(cnt[k], 1)
        gui = v['GRBM_GUI_ACTIVE'] / n / 8     # cycles per launch (8 XCD instances summed)
        print(f"{k:85s} n={n:3d} cyc/launch={gui:10.0f} mfma_busy%={100*v['SQ_VALU_MFMA_BUSY_CYCLES']/n/(1024*gui):5.1f} insts_mfma={v['SQ_INSTS_MFMA']/n:.3e} insts_valu={v['SQ_INSTS_VALU']/n:.3e}")
PY
