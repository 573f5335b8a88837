#!/bin/bash
# same-box A/B of the cost-volume build INSIDE bench.py's stage: the round-5 single-stage strip kernel (PGRF_CV_PIPE=0 PGRF_CV_TILE_W=64)
# against the shipped pipelined instance, alternating, three rounds:   bash tools/cv_bench_ab.sh
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
for v in r05-kernel shipped; do
  unset PGRF_CV_PIPE PGRF_CV_TILE_W
  if [ $v = r05-kernel ]; then export PGRF_CV_PIPE=0 PGRF_CV_TILE_W=64; fi
  for reg in costregnet unet2; do
  echo -n "$v $reg: "
  python bench.py --with-cost-volume --reg $reg --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['cost_volume_stage']; print('stage', round(c['ms_per_step'],3), 'ms; build', round(c['roofline_build']['kernel_ms']*1e3,1), 'us per view =', round(c['roofline_build']['frac']*100,1), '% of 8 TB/s; regulariser', round(c['roofline']['kernel_ms'],3), 'ms per view; frame', round(d['ms_per_step'],2), 'ms')"
  done
done; done
