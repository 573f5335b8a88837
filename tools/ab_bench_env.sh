#!/bin/bash
# A/B of an environment knob on the bench line:  bash tools/ab_bench_env.sh VAR "v1|v2|..." [bench args]   (values separated by |)
VAR=$1; IFS='|' read -ra VALS <<< "$2"; shift 2
for rep in 1 2; do for v in "${VALS[@]}"; do
  echo -n "$VAR='$v': "; env "$VAR=$v" python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2),'ms', round(d['roofline']['frac'],3))"
done; done
