#!/bin/bash
# A/B of the deep-ring policy on FlowNet2 b4 / FlowNetC b8 (same box, alternating)
run() { # label, env...
  local label=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --regions 3 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$label', d['ms_per_step'], d['epe_vs_oracle_fixture_px'], d['roofline']['frac'])"
}
for rep in 1 2; do
  run "F2 base        " FN2_DEEP_MAX=-1
  run "F2 deep 128-383" FN2_DEEP_MIN=128 FN2_DEEP_MAX=383
  run "F2 deep 96-383 " FN2_DEEP_MIN=96 FN2_DEEP_MAX=383
  run "F2 deep 24-383 " FN2_DEEP_MIN=24 FN2_DEEP_MAX=383
  run "F2 deep 128-512" FN2_DEEP_MIN=128 FN2_DEEP_MAX=512
done
EXTRA="--model FlowNetC"
run "C8 base        " FN2_DEEP_MAX=-1
run "C8 deep 128-383" FN2_DEEP_MIN=128 FN2_DEEP_MAX=383
run "C8 deep 24-383 " FN2_DEEP_MIN=24 FN2_DEEP_MAX=383
