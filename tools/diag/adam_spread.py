"""Spread of Adam's moments between the captured and the eager train step (tests/test_gpu_train.py:
test_captured_train_step_equals_eager): per dtype, max over layers of max|v_graph - v_eager| / max|v_eager| and the
same for m -- the number the test's tolerance is set from."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-tf_amd"), os.path.join(ROOT, "tests")]
from test_gpu_train import data  # noqa: E402
from src import weights as W  # noqa: E402
from src.trainer import FlowNetSTrainer  # noqa: E402

for dtype in ("f32", "f16x2"):
    wts = W.init_weights("FlowNetS", 6)
    batches = [data(2, 128, 128, 20 + i) for i in range(3)]
    os.environ["FN2_TRAIN_GRAPH"] = "0"
    eager = FlowNetSTrainer(wts, 2, 128, 128, dtype=dtype)
    for b in batches:
        eager.train_step(*b)
    os.environ["FN2_TRAIN_GRAPH"] = "1"
    graph = FlowNetSTrainer(wts, 2, 128, 128, dtype=dtype)
    for b in batches:
        graph.train_step(*b)
    worst_v, worst_m = (0, ""), (0, "")
    for pe, pg in zip(eager.params, graph.params):
        ve, vg = pe["v"].cpu().numpy(), pg["v"].cpu().numpy()
        me, mg = pe["m"].cpu().numpy(), pg["m"].cpu().numpy()
        rv = np.abs(vg - ve).max() / max(np.abs(ve).max(), 1e-30)
        rm = np.abs(mg - me).max() / max(np.abs(me).max(), 1e-30)
        worst_v, worst_m = max(worst_v, (rv, pe["name"])), max(worst_m, (rm, pe["name"]))
    print(dtype, "v spread %.3e (%s)   m spread %.3e (%s)" % (worst_v + worst_m), flush=True)
