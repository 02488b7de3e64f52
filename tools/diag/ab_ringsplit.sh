#!/bin/bash
run() { # label, env...
  local label=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --regions 3 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$label', d['ms_per_step'], d['epe_vs_oracle_fixture_px'], d['roofline']['frac'])"
}
for rep in 1 2; do
  run "F2 base              "
  run "F2 ringsplit>=96     " FN2_RING_SPLIT=96
  run "F2 ringsplit>=96 s384" FN2_RING_SPLIT=96 FN2_RING_SPLIT_SLOTS=384
  run "F2 ringsplit>=192    " FN2_RING_SPLIT=192
  run "F2 slots 512         " FN2_SPLIT_SLOTS=512
  run "F2 slots 768         " FN2_SPLIT_SLOTS=768
done
