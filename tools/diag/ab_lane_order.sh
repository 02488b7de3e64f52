#!/bin/bash
# A/B of the capture order of the lanes (FN2_LANE_ORDER), same box, alternating runs
for rep in 1 2; do
  for m in list fair lanes; do
    FN2_LANE_ORDER=$m python bench.py --no-extra --no-cpu-baseline --regions 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('FlowNet2 b4 $m', d['ms_per_step'], d['epe_vs_oracle_fixture_px'])"
  done
done
for m in list fair lanes; do
  FN2_LANE_ORDER=$m python bench.py --model FlowNetC --no-extra --no-cpu-baseline --regions 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('FlowNetC b8 $m', d['ms_per_step'], d['epe_vs_oracle_fixture_px'])"
done
