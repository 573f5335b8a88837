#!/bin/bash
# SQ counters of the Winograd kernel over tools/check_wino.py time:  bash tools/pmc_wino.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_wino
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/tools/check_wino.py time > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/tools/check_wino.py time > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/p3 -- python3 $GRAFT_REPO_ROOT/tools/check_wino.py time > $OUT/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'wino' not in row['Kernel_Name']: continue
        k = ('wino16' if 'wino16' in row['Kernel_Name'] else 'wino32') + ' grid=' + row['Grid_Size']
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[k][row['Counter_Name']] += 1
for k, v in sorted(agg.items()):
    g = lambda c: v.get(c, 0.0) / max(cnt[k][c], 1)
    gui = g('GRBM_GUI_ACTIVE') / 8
    wc = max(g('SQ_WAVE_CYCLES'), 1)
    print(k, 'n=', cnt[k]['SQ_INSTS_MFMA'], f"cyc={gui:.0f} mfma_busy%={100*g('SQ_VALU_MFMA_BUSY_CYCLES')/(1024*gui):.1f}")
    for c in sorted(v):
        print(f'    {c:28s} {g(c):.4e}   per MFMA {g(c)/max(g("SQ_INSTS_MFMA"),1):8.3f}   /wave_cycles {g(c)/wc:6.3f}')
PY
