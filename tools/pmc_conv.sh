#!/bin/bash
# SQ counters of the conv3d kernels in one regulariser run, grouped by (kernel, grid): bash tools/pmc_conv.sh TAG [unet2|costregnet] [h w]
cd /tmp && export TMPDIR=/tmp
TAG=$1; NET=${2:-unet2}; H=${3:-128}; W=${4:-256}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcv_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/tools/time_cost_volume.py $H $W $NET > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/tools/time_cost_volume.py $H $W $NET > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/p3 -- python3 $GRAFT_REPO_ROOT/tools/time_cost_volume.py $H $W $NET > $OUT/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'conv3d' not in row['Kernel_Name']: continue
        k = row['Kernel_Name'][11:50] + ' grid=' + row['Grid_Size']
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[k][row['Counter_Name']] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))[:10]:
    n = max(cnt[k]['SQ_INSTS_MFMA'], 1)
    g = lambda c: v.get(c, 0.0) / max(cnt[k][c], 1)
    gui = g('GRBM_GUI_ACTIVE') / 8
    wc = g('SQ_WAVE_CYCLES')
    print(f"{k:62s} n={n:3d} cyc={gui:9.0f} mfma_busy%={100*g('SQ_VALU_MFMA_BUSY_CYCLES')/(1024*gui):5.1f} valu/mfma={g('SQ_INSTS_VALU')/max(g('SQ_INSTS_MFMA'),1):5.2f} "
          f"wait_any%={100*g('SQ_WAIT_ANY')/wc:5.1f} wait_inst%={100*g('SQ_WAIT_INST_ANY')/wc:5.1f} active%={100*g('SQ_ACTIVE_INST_ANY')/wc:5.1f} vmem_rd/mfma={g('SQ_INSTS_VMEM_RD')/max(g('SQ_INSTS_MFMA'),1):5.3f} salu/mfma={g('SQ_INSTS_SALU')/max(g('SQ_INSTS_MFMA'),1):5.2f} "
          f"wait_lds%={100*g('SQ_WAIT_INST_LDS')/wc:5.1f} lds_conflict/lds_active={g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_LDS_IDX_ACTIVE'),1):5.3f} lds/mfma={g('SQ_INSTS_LDS')/max(g('SQ_INSTS_MFMA'),1):5.3f}")
PY
