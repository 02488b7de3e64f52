#!/bin/bash
run() { # label, env...
  local label=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --regions 3 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$label', d['ms_per_step'], d['epe_vs_oracle_fixture_px'], d['roofline']['kernel'][-30:], d['roofline']['frac'])"
}
for rep in 1 2; do
  run "F2 wreg 2stage only      " FN2_WREG_RING=0
  run "F2 wreg +ringbase as ring" FN2_WREG_RING=1 FN2_WREG_KERNELS="1, false>"
  run "F2 wreg +ringbase 2stage " FN2_WREG_RING=0 FN2_WREG_KERNELS="1, false>"
  run "F2 wreg ring everywhere  " FN2_WREG_RING=2 FN2_WREG_KERNELS="1, false>"
  run "F2 wreg ring on 2stage   " FN2_WREG_RING=2
done
EXTRA="--model FlowNetC"
run "C8 wreg 2stage only      " FN2_WREG_RING=0
run "C8 wreg +ringbase as ring" FN2_WREG_RING=1 FN2_WREG_KERNELS="1, false>"
run "C8 wreg ring everywhere  " FN2_WREG_RING=2 FN2_WREG_KERNELS="1, false>"
