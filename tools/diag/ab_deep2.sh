#!/bin/bash
run() { # label, env...
  local label=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --regions 3 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$label', d['ms_per_step'], d['epe_vs_oracle_fixture_px'], d['roofline']['frac'])"
}
for rep in 1 2; do
  run "F2 base          " FN2_DEEP_MAX=-1
  run "F2 deep 0-95     " FN2_DEEP_MIN=0 FN2_DEEP_MAX=95
  run "F2 deep 0-95 s384" FN2_DEEP_MIN=0 FN2_DEEP_MAX=95 FN2_DEEP_SLOTS=384
  run "F2 deep 0-95 s512" FN2_DEEP_MIN=0 FN2_DEEP_MAX=95 FN2_DEEP_SLOTS=512
  run "F2 deep 0-127    " FN2_DEEP_MIN=0 FN2_DEEP_MAX=127
done
EXTRA="--model FlowNetC"
run "C8 base        " FN2_DEEP_MAX=-1
run "C8 deep 0-95   " FN2_DEEP_MIN=0 FN2_DEEP_MAX=95
run "C8 deep 0-127  " FN2_DEEP_MIN=0 FN2_DEEP_MAX=127
run "C8 deep 0-191  " FN2_DEEP_MIN=0 FN2_DEEP_MAX=191
