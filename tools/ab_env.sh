#!/bin/bash
# A/B of an environment knob on the regulariser timing:  bash tools/ab_env.sh VAR "v1 v2" [net h w]
VAR=$1; VALS=$2; NET=${3:-unet2}; H=${4:-128}; W=${5:-256}
for rep in 1 2; do for v in $VALS; do
  echo "== $VAR=$v: $(env $VAR=$v python3 $GRAFT_REPO_ROOT/tools/time_cost_volume.py $H $W $NET 2>&1 | grep -E 'UNet2|CostRegNet')"
done; done
