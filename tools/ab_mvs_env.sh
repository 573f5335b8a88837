#!/bin/bash
# A/B of an environment knob on the MVS stage and the per-frame encoders:  bash tools/ab_mvs_env.sh VAR "v1|v2|..."
VAR=$1; IFS='|' read -ra VALS <<< "$2"
for rep in 1 2; do for v in "${VALS[@]}"; do
  m=$(env "$VAR=$v" python3 $GRAFT_REPO_ROOT/tools/time_mvs.py 2>&1 | grep -E "2-D nets on HIP, replayed|HIP Equi feature net|HIP res_net" | sed 's/.*: //' | tr '\n' ' ')
  e=$(env "$VAR=$v" python3 $GRAFT_REPO_ROOT/tools/time_encoders.py 2>&1 | grep "encoders on HIP, hipGraph" | sed 's/.*: //')
  echo "$VAR='$v': mvs stage | equi | res_net = $m | encoders $e"
done; done
