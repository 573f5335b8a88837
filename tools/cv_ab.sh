#!/bin/bash
# kernel-trace A/B of cost-volume library variants: bash tools/cv_ab.sh base cvg cvs ...
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset PANOGRF_HIP_LIB; else export PANOGRF_HIP_LIB=$GRAFT_REPO_ROOT/panogrf_amd/libpanogrf_hip_$v.so PGRF_ALLOW_ABLATION_LIB=1; fi
  OUT=$GRAFT_REPO_ROOT/gpurun_out/cvab_$v
  rm -rf $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/time_cost_volume.py 128 256 costregnet > $OUT.log 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'cost_volume' in r['Name']: print('$v:', round(float(r['AverageNs'])/1000,1), 'us avg,', r['Calls'], 'calls')
"
  rm -rf $OUT
done
