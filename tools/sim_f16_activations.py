#!/usr/bin/env python
"""Accuracy of a "2-MFMA" operand mode (activations stored as plain fp16, weights exact) BEFORE building it: the NumPy
oracle with every activation rounded to fp16 between layers (flow heads stay fp32, as in the engine) against the plain
oracle, seeded weights.  CPU only:   python tools/sim_f16_activations.py
Result: FlowNetS 7.8e-4, FlowNetC 7.0e-4, FlowNet2 1.6e-4 px -- under the 1e-3 px bar without
margin on S and C, so the mode was not built."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'flownet2-tf_amd')]
import numpy as np
from oracle import nn as refnn, models as refm
from src import weights as W

def images(n, h, w, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (n, h, w, 3)).astype(np.float32) / 255.0
    b = np.roll(a, (3, -5), (1, 2)) + rng.uniform(-4, 4, a.shape).astype(np.float32) / 255.0
    return a, np.clip(b, 0, 1).astype(np.float32)

orig_conv, orig_deconv = refnn.conv2d, refnn.conv2d_transpose
Q = {"on": False}
def q16(x):
    return x.astype(np.float16).astype(x.dtype) if Q["on"] else x
def conv2d(x, w, b=None, **kw):
    y = orig_conv(q16(x), w, b, **kw)
    return y if w.shape[3] == 2 else q16(y)
def deconv(x, w, *a, **kw):
    return q16(orig_deconv(q16(x), w, *a, **kw))
refnn.conv2d, refnn.conv2d_transpose = conv2d, deconv
for m in (refm,):
    for name in ("conv2d", "conv2d_transpose"):
        if hasattr(m, name): setattr(m, name, getattr(refnn, name))

def epe(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt((d * d).sum(-1)).mean())

for model, (n, h, w) in (("FlowNetS", (1, 128, 192)), ("FlowNetC", (1, 128, 192)), ("FlowNet2", (1, 64, 128))):
    wts = W.init_weights(model, 1234)
    a, b = images(n, h, w, 3)
    Q["on"] = False
    t0 = time.time(); want = refm.MODELS[model](wts, {"input_a": a, "input_b": b})["flow"]
    Q["on"] = True
    got = refm.MODELS[model](wts, {"input_a": a, "input_b": b})["flow"]
    print(model, (n, h, w), "fp16 activations, exact weights: mean EPE %.3e px, mean |flow| %.3f  (%.0f s)" % (epe(got, want), float(np.sqrt((want**2).sum(-1)).mean()), time.time()-t0), flush=True)
