#!/usr/bin/env python
"""Idle time inside captured forward steps, from a rocprofv3 --kernel-trace CSV:
    rocprofv3 --kernel-trace -d out --output-format csv -- python3 bench.py --no-extra --no-cpu-baseline --regions 1
    python tools/graph_gaps.py out/*/*kernel_trace.csv
Splits the trace at the first kernel of a step (the name that opens the graph), then reports per step: wall (first
start .. last end), the union of the kernel intervals (time with at least one kernel running), the idle remainder and
the number of launches."""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
first = sys.argv[2] if len(sys.argv) > 2 else None
if first is None:  # the most frequent "first kernel after a long quiet period" is not robust: take the pack kernel
    first = next(n for _, _, n in ev if "pack_image" in n)
starts = [i for i, e in enumerate(ev) if first in e[2]]
# steps of the timed region: the last 30 occurrences that are one graph replay apart
steps = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = ev[a:b]
    if len(seg) < 50:
        continue
    t0, t1 = seg[0][0], max(e[1] for e in seg)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    steps.append((t1 - t0, busy, sum(e - s for s, e, _ in seg), len(seg)))
steps = np.array(steps[-30:], float)
print("steps %d | wall %.3f ms | >=1 kernel running %.3f ms | idle %.3f ms | sum of kernel durations %.3f ms | launches %d" % (
    len(steps), steps[:, 0].mean() / 1e6, steps[:, 1].mean() / 1e6, (steps[:, 0] - steps[:, 1]).mean() / 1e6,
    steps[:, 2].mean() / 1e6, int(steps[:, 3].mean())))
