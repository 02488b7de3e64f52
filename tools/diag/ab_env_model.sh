#!/bin/bash
# A/B of one environment variable on another workload: tools/diag/ab_env_model.sh "<bench args>" VAR val1 val2 ...
args=$1; var=$2; shift 2
for rep in 1 2; do
  for v in "$@"; do
    env $var=$v python bench.py $args --no-extra --no-cpu-baseline --regions 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$var=$v', d['ms_per_step'], d['epe_vs_oracle_fixture_px'])"
  done
done
