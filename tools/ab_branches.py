#!/usr/bin/env python
"""A/B of the hipGraph with and without parallel branches (FlowNet2: FlowNetSD beside the C -> S -> S chain).
Same process, interleaved rounds; checks that the replayed flow equals the eager one bit for bit.

  python tools/ab_branches.py [--batch 4 --rounds 5 --steps 30]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-tf_amd")]
from src import weights as W  # noqa: E402
from src.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="FlowNet2")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--masks", default="0,3,7,23", help="FN2_LANE_MASK values to compare (0 = no lanes)")
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    ia = rng.random((a.batch, a.height, a.width, 3), dtype=np.float32)
    ib = np.roll(ia, (3, -5), (1, 2))
    wts = W.init_weights(a.model, 1234)
    engs = {}
    masks = [int(m) for m in a.masks.split(",")]
    for br in masks:
        os.environ["FN2_BRANCHES"] = "1" if br else "0"
        os.environ["FN2_LANE_MASK"] = str(br)
        e = Engine(a.model, wts, a.batch, a.height, a.width, "f16x2")
        eager = e(ia, ib)["flow"].clone()
        e.capture()
        e.launch()
        torch.cuda.synchronize()
        assert torch.equal(e.outputs["flow"], eager), "graph replay differs from eager (branches=%d)" % br
        engs[br] = e
    assert all(torch.equal(engs[masks[0]].outputs["flow"], e.outputs["flow"]) for e in engs.values())
    res = {m: [] for m in masks}
    for _ in range(a.rounds):
        for br, e in engs.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                e.launch()
            torch.cuda.synchronize()
            res[br].append((time.perf_counter() - t0) / a.steps * 1e3)
    for br in masks:
        print("lane mask=%d  ms/step median %.4f  min %.4f  max %.4f" % (br, np.median(res[br]), min(res[br]), max(res[br])))


if __name__ == "__main__":
    main()
