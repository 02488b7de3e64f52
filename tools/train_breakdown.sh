#!/bin/bash
# tools/train_breakdown.sh [bench args]   e.g. --train-dtype f16x2
python bench.py --mode train --steps 10 --warmup 3 --per-layer "$@" 2>gpurun_out/train_layers.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['dtype'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step']): print(k, v)"
