"""Diagnostic: error distribution of the batch-8 train-plan gradients against the committed fixture, and against the
oracle re-run on this box with the LeakyReLU branches the device took (tests/test_gpu_train.py: device_signs)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-tf_amd"), os.path.join(ROOT, "tests")]
import bench
from src import weights as W
from src.trainer import FlowNetSTrainer
from oracle import train as reft
from test_gpu_train import device_signs

dtype = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
g = np.load(os.path.join(ROOT, "tests/golden/plan_flownets_train_b8_384x512.npz"))
a, b = bench.synth_pairs(8, 384, 512, 0)
gt = bench.synth_gt(8, 384, 512, 0)
wts = W.init_weights("FlowNetS", 1234)
tr = FlowNetSTrainer(wts, 8, 384, 512, dtype=dtype)
loss = float(tr.forward_backward(a, b, gt).item())
print("loss", loss, float(g["loss"]))
got = {p["name"]: tr._to_reference_layout(p, p["g"]) / np.float32(tr.loss_scale) for p in tr.params}
for name, gv in got.items():
    if name in g.files:
        want, gotv, scale = g[name].reshape(-1), gv.reshape(-1), np.abs(g[name]).max()
    else:
        idx = g[name + "#idx"]; want = g[name + "#val"]; gotv = gv.reshape(-1)[idx]; scale = float(g[name + "#max"])
    e = np.abs(gotv - want) / scale
    print("fixture %-36s max %.2e p99 %.2e p50 %.2e  n>2e-5: %d / %d" % (name, e.max(), np.percentile(e, 99), np.median(e), (e > 2e-5).sum(), e.size))
if "--signs" in sys.argv:
    signs = device_signs(tr)
    t0 = time.time()
    grads = None
    for i in range(8):
        sg = {k: v[i:i + 1] for k, v in signs.items()}
        l, gr, _ = reft.flownet_s_loss_and_grads(wts, a[i:i + 1], b[i:i + 1], gt[i:i + 1], signs=sg)
        grads = {k: v / 8 for k, v in gr.items()} if grads is None else {k: grads[k] + v / 8 for k, v in gr.items()}
    print("oracle with device signs: %.0f s" % (time.time() - t0))
    for name, gv in got.items():
        want = grads[name]
        e = np.abs(gv.reshape(-1) - want.reshape(-1)) / (np.abs(want).max() + 1e-30)
        print("signs   %-36s max %.2e p99 %.2e" % (name, e.max(), np.percentile(e, 99)))
