#!/bin/bash
# End-to-end A/B of one environment knob of the library on ONE box:
#   tools/ab_env.sh FN2_RING_MAX "0 512" --model FlowNet2 --batch 4
var=$1; vals=$2; shift 2
for rep in 1 2; do
  for v in $vals; do
    export $var=$v
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('$var', os.environ['$var'], d['ms_per_step'], d['value'])"
  done
done
