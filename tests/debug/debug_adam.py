"""Debug helper (GPU): Adam steps of the trainer vs the oracle, per layer statistics."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-tf_amd"), os.path.join(ROOT, "tests")]
from oracle import train as reft
from src import weights as W
from src.trainer import FlowNetSTrainer
from test_gpu_train import data, device_signs, packed_grad

dtype = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
wts = W.init_weights("FlowNetS", 6)
a, b, gt = data(1, 128, 128, 2)
tr = FlowNetSTrainer(wts, 1, 128, 128, dtype=dtype)
cur = {k: np.asarray(v, np.float64) for k, v in wts.items()}
mom = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in cur.items()}
l2 = tr.schedule["l2_regularization"]
prev = {k: v.copy() for k, v in cur.items()}
for step in (1, 2):
    loss = float(tr.forward_backward(a, b, gt).item())
    signs = device_signs(tr)
    tr.apply_gradients()
    want_loss, grads, _ = reft.flownet_s_loss_and_grads(cur, a, b, gt, l2=l2, signs=signs)
    print("step", step, "loss", loss, "oracle (incl. l2)", want_loss)
    for k in cur:
        prev[k] = cur[k].copy()
        cur[k], m, v = reft.adam_update(cur[k], grads[k], mom[k][0], mom[k][1], step)
        mom[k] = (m, v)
    for rec in tr.eng.layers[:4] + tr.eng.layers[-3:]:
        name = f"{rec['scope']}/{rec['name']}/weights"
        got = rec["master"].cpu().numpy().reshape(-1)
        pk = (lambda x: x.reshape(-1)) if rec["kind"] == "upflow" else (lambda x: packed_grad(rec, x).reshape(-1))
        want, w0 = pk(cur[name].astype(np.float32)), pk(prev[name].astype(np.float32))
        mv_g, mv_w = got - w0, want - w0
        bad = np.abs(mv_g - mv_w) > 0.05 * np.abs(mv_w).max()
        print("  %-26s max move %.3e  max diff %.3e  frac bad %.2e" % (rec["name"], np.abs(mv_w).max(), np.abs(mv_g - mv_w).max(), bad.mean()))
