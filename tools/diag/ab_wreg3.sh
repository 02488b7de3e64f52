#!/bin/bash
run() { # label, env...
  local label=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --regions 3 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$label', d['ms_per_step'], d['epe_vs_oracle_fixture_px'], d['roofline']['kernel'][-30:], d['roofline']['frac'])"
}
for rep in 1 2; do
  run "F2 wreg (2stage+ring base)    " FN2_WREG_KERNELS="1, false>"
  run "F2 wreg all 128-cout incl KG   " FN2_WREG_KERNELS="false>"
  run "F2 wreg all, slots 1024        " FN2_WREG_KERNELS="false>" FN2_SPLIT_SLOTS=1024
  run "F2 wreg all, slots 1024 ring2  " FN2_WREG_KERNELS="false>" FN2_SPLIT_SLOTS=1024 FN2_WREG_RING=2
done
