#!/bin/bash
run() { # label, env...
  local label=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --regions 3 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$label', d['ms_per_step'], d['epe_vs_oracle_fixture_px'], d['roofline']['kernel'][-30:], d['roofline']['frac'])"
}
for rep in 1 2; do
  run "F2 wreg 128 only   " FN2_WREG_TILE_MIN=128
  run "F2 wreg 128 + 64   " FN2_WREG_TILE_MIN=64
  run "F2 no wreg         " FN2_WREG=0
done
EXTRA="--model FlowNetC"
run "C8 wreg 128 only   " FN2_WREG_TILE_MIN=128
run "C8 wreg 128 + 64   " FN2_WREG_TILE_MIN=64
run "C8 no wreg         " FN2_WREG=0
