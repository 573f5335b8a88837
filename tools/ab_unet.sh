#!/bin/bash
for v in "$@"; do
  if [ "$v" = base ]; then unset PANOGRF_HIP_LIB; else export PANOGRF_HIP_LIB=$GRAFT_REPO_ROOT/panogrf_amd/libpanogrf_hip_$v.so PGRF_ALLOW_ABLATION_LIB=1; fi
  echo -n "$v: "; python3 $GRAFT_REPO_ROOT/tools/time_cost_volume.py ${HW:-128 256} unet2 2>&1 | grep UNet2
done
