#!/bin/bash
# A/B timing of library variants built by tools/build_variant_src.sh:  bash tools/ab_conv.sh "v1 v2 ..." [net h w]  ("main" = the in-tree library)
VARS=${1:-main}; NET=${2:-unet2}; H=${3:-128}; W=${4:-256}
for rep in 1 2; do
for v in $VARS; do
  if [ $v = main ]; then unset PANOGRF_HIP_LIB; else export PANOGRF_HIP_LIB=$GRAFT_REPO_ROOT/panogrf_amd/libpanogrf_hip_$v.so PGRF_ALLOW_ABLATION_LIB=1; fi
  echo "== $v: $(python3 $GRAFT_REPO_ROOT/tools/time_cost_volume.py $H $W $NET 2>&1 | grep -E 'UNet2|CostRegNet')"
done; done
