"""Experiment: FlowNet2 batch 4 as ONE captured plan vs TWO batch-2 plans replayed concurrently on two streams vs FOUR
batch-1 plans (more independent chains to fill the CUs that small grids / serial tiny kernels leave idle)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-tf_amd")]
import bench
from src import weights as W
from src.engine import Engine

model = sys.argv[1] if len(sys.argv) > 1 else "FlowNet2"
total = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wts = W.init_weights(model, 1234)
a, b = bench.synth_pairs(total, 384, 512, 0)

def run(parts, steps=30, reps=5):
    n = total // parts
    engs, streams = [], []
    for i in range(parts):
        e = Engine(model, wts, n, 384, 512, "f16x2")
        e.set_inputs(a[i * n:(i + 1) * n], b[i * n:(i + 1) * n])
        torch.cuda.synchronize()
        e.capture()
        engs.append(e)
        streams.append(torch.cuda.Stream())
    def step():
        cur = torch.cuda.current_stream()
        for e, s in zip(engs, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                e.launch()
        for s in streams:
            cur.wait_stream(s)
    for _ in range(5):
        step()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / steps * 1e3)
    return float(np.median(ts))

for parts in (1, 2, 4):
    if total % parts == 0:
        print("%s total batch %d as %d concurrent plan(s) of batch %d: %.3f ms per step" % (model, total, parts, total // parts, run(parts)), flush=True)
