#!/bin/bash
# A/B of library variants: tools/ab_variants.sh name1 name2 ...   ("base" = panogrf_amd/libpanogrf_hip.so)
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset PANOGRF_HIP_LIB; else export PANOGRF_HIP_LIB=$GRAFT_REPO_ROOT/panogrf_amd/libpanogrf_hip_$v.so PGRF_ALLOW_ABLATION_LIB=1; fi
  echo -n "$v: "; python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2),'ms', round(d['roofline']['frac'],3))"
done; done
