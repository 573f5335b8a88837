#!/bin/bash
# A/B of library variants on the training step: tools/ab_train_variants.sh base name1 ...   ("base" = panogrf_amd/libpanogrf_hip.so)
R=${GRAFT_REPO_ROOT:?run through gpurun}
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset PANOGRF_HIP_LIB; else export PANOGRF_HIP_LIB=$R/panogrf_amd/libpanogrf_hip_$v.so PGRF_ALLOW_ABLATION_LIB=1; fi
  echo -n "$v: "; python bench.py --train --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3),'ms', round(d['roofline']['frac'],3))"
done; done
