#!/bin/bash
# A/B of one environment variable on the FlowNet2 b4 step: tools/diag/ab_env.sh VAR val1 val2 ...
var=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    env $var=$v python bench.py --no-extra --no-cpu-baseline --regions 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$var=$v', d['ms_per_step'], d['epe_vs_oracle_fixture_px'])"
  done
done
