#!/bin/bash
# End-to-end A/B of FN2_CONV_DBG settings on ONE box: tools/ab_e2e.sh "0 64 32" [bench args...]
modes=$1; shift
for rep in 1 2; do
  for dbg in $modes; do
    export FN2_CONV_DBG=$dbg
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('dbg', os.environ['FN2_CONV_DBG'], d['ms_per_step'], d['value'])"
  done
done
