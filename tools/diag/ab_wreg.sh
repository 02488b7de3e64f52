#!/bin/bash
run() { # label, env...
  local label=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --regions 3 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$label', d['ms_per_step'], d['epe_vs_oracle_fixture_px'], d['roofline']['kernel'][-40:], d['roofline']['frac'])"
}
for rep in 1 2; do
  run "F2 layout1 " FN2_WREG=0
  run "F2 wreg    " FN2_WREG=1
  run "F2 wreg+ring" FN2_WREG=1 FN2_WREG_KERNELS="1, false>"
done
EXTRA="--model FlowNetC"
run "C8 layout1 " FN2_WREG=0
run "C8 wreg    " FN2_WREG=1
run "C8 wreg+ring" FN2_WREG=1 FN2_WREG_KERNELS="1, false>"
