"""Robustness sweep (GPU): every model at odd batches / sizes runs and stays finite; FlowNetS also against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-tf_amd")]
from src import weights as W
from src.engine import Engine
from oracle import models as refm

rng = np.random.default_rng(0)
for model, n, h, w, dt in [("FlowNet2", 1, 64, 64, "f16x2"), ("FlowNet2", 3, 128, 192, "f16x2"), ("FlowNetC", 3, 384, 512, "f16x2"),
                           ("FlowNetCSS", 5, 64, 128, "bf16"), ("FlowNetSD", 7, 64, 64, "f16"), ("FlowNetS", 2, 448, 1024, "f16x2"),
                           ("FlowNetC", 1, 64, 64, "f32"), ("FlowNetS_interp", 2, 128, 128, "f16x2")]:
    wts = W.init_weights(model, 3)
    a = rng.random((n, h, w, 3), dtype=np.float32)
    b = np.roll(a, (1, -2), (1, 2))
    eng = Engine(model, wts, n, h, w, dt)
    out = eng(a, b)["flow"].float().cpu().numpy()
    msg = "%-16s n=%d %dx%d %-6s finite=%s |flow| mean %.3f" % (model, n, h, w, dt, np.isfinite(out).all(), np.abs(out).mean())
    if model in ("FlowNetS", "FlowNetC") and h * w <= 64 * 64 * 200 and dt in ("f32", "f16x2"):
        want = refm.MODELS[model](wts, {"input_a": a, "input_b": b})["flow"]
        msg += "  EPE vs oracle %.2e" % float(np.sqrt(((out - want) ** 2).sum(-1)).mean())
    print(msg, flush=True)
    assert np.isfinite(out).all()
    eng.capture()
    eng.launch(); eng.launch()
    out2 = eng.outputs["flow"].float().cpu().numpy()
    assert np.array_equal(out, out2), "graph replay differs"
print("sweep ok")
