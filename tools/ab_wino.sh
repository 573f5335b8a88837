#!/bin/bash
# Winograd layer timings (tools/check_wino.py time) for library variants:  bash tools/ab_wino.sh "main v1 v2" [line numbers of the table]
VARS=${1:-main}; LINES=${2:-1,8p}
for v in $VARS; do
  if [ $v = main ]; then unset PANOGRF_HIP_LIB; else export PANOGRF_HIP_LIB=$GRAFT_REPO_ROOT/panogrf_amd/libpanogrf_hip_$v.so PGRF_ALLOW_ABLATION_LIB=1; fi
  echo "== $v"; python3 $GRAFT_REPO_ROOT/tools/check_wino.py time 2>&1 | sed -n $LINES
done
