#!/usr/bin/env python
"""Timeline of ONE captured step from a rocprofv3 --kernel-trace CSV (see tools/graph_gaps.py for the command):
per kernel start offset, duration, queue and short name, then the time attributed to every kernel family when each
instant of the step is split evenly among the kernels running in it (what the step's WALL time is made of, lanes
included -- the per-launch event times of bench.py are eager and sequential)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
first = next(n for _, _, n, _ in ev if "pack_image" in n or "u8_to_f32" in n)  # the kernel that opens a step
starts = [i for i, e in enumerate(ev) if e[2] == first and (i == 0 or ev[i - 1][2] != first)]
segs = [ev[a:b] for a, b in zip(starts[:-1], starts[1:]) if b - a > 50]
seg = segs[which]
t0 = seg[0][0]
t1 = max(e[1] for e in seg)


def short(n):
    n = n.replace("void fn2::", "").replace("fn2::", "")
    return n[:n.index("(")] if "(" in n else n


print("step wall %.1f us, %d kernels" % ((t1 - t0) / 1e3, len(seg)))
if "-v" in sys.argv:
    for s, e, n, q in seg:
        print("%9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short(n)[:90]))
# even split of every instant among the running kernels
pts = sorted({p for s, e, _, _ in seg for p in (s, e)})
share = defaultdict(float)
alone = defaultdict(float)
idle = 0.0
for a, b in zip(pts[:-1], pts[1:]):
    run = [n for s, e, n, _ in seg if s <= a and e >= b]
    if not run:
        idle += b - a
        continue
    for n in run:
        share[short(n)] += (b - a) / len(run)
    if len(run) == 1:
        alone[short(run[0])] += b - a
tot = sum(share.values())
print("idle %.1f us" % (idle / 1e3))
dur = defaultdict(float)
cnt = defaultdict(int)
for s, e, n, _ in seg:
    dur[short(n)] += e - s
    cnt[short(n)] += 1
print("%-86s %9s %9s %9s %5s" % ("kernel", "share us", "alone us", "sum us", "n"))
for n, v in sorted(share.items(), key=lambda kv: -kv[1]):
    print("%-86s %9.1f %9.1f %9.1f %5d" % (n[:86], v / 1e3, alone[n] / 1e3, dur[n] / 1e3, cnt[n]))
