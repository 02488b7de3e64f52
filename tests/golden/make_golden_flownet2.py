"""Generate tests/golden/flownet2_{384x512,448x1024}_golden.npz: BASELINE configs 3 and 5 -- one pair of the
FlowNet2 full stack (C -> S -> S, SD, fusion; flownet2.py:18-105) at 512x384 and at the Sintel shape (a 436 x 1024
pair zero-padded to 448 x 1024 as Net.adapt_x does, net.py:373-388) through the NumPy oracle with seeded weights
(src.weights.init_weights('FlowNet2', 1234) -- the fusion net's four transposed-conv biases included and non-zero).

Stored per shape: 4096 probe pixels of the final `flow` and of the intermediate FlowNetC / CS / CSS / SD flows
(so a GPU test localises a deviation to a sub-network), the probe coordinates, and mean |flow|.  The oracle is
this build's restatement: these fixtures are a REGRESSION pin of the oracle and a device-free target for the
full-size GPU tests in the bench dtype (the oracle takes minutes at these sizes on 8 cores), not a pin against the
reference.

    python tests/golden/make_golden_flownet2.py            # both shapes (~10 minutes on 8 cores)
    python tests/golden/make_golden_flownet2.py 384x512
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-tf_amd")]

SEED_W = 1234
SHAPES = {"384x512": (384, 512, 384), "448x1024": (448, 1024, 436)}  # padded H, W, rows holding image data


def inputs(name):
    """SURVEY.md section 8d synthetic pair (smoothed uint8 noise, second = first rolled by (3,-5) + noise), [0,1]
    floats; rows past the image's own height are zero (adapt_x's bottom padding)."""
    H, W, rows = SHAPES[name]
    rng = np.random.default_rng(77)
    a = rng.integers(0, 256, (1, rows, W, 3)).astype(np.float32)
    a = (a + np.roll(a, 1, 1) + np.roll(a, 1, 2) + np.roll(a, (1, 1), (1, 2))) / 4
    b = np.clip(np.roll(a, (3, -5), (1, 2)) + rng.uniform(-4, 4, a.shape), 0, 255)
    pad = [(0, 0), (0, H - rows), (0, 0), (0, 0)]
    return np.pad(a / 255, pad).astype(np.float32), np.pad(b / 255, pad).astype(np.float32)


def probes(name):
    H, W, _ = SHAPES[name]
    rng = np.random.default_rng(4242)
    return rng.integers(0, H, 4096), rng.integers(0, W, 4096)


def main(names):
    from oracle import models as refm
    from src import weights as W
    wts = W.init_weights("FlowNet2", SEED_W)
    for name in names:
        a, b = inputs(name)
        inp = {"input_a": a, "input_b": b}
        t0 = time.time()
        # flownet2.py:22-23 unrolled one level so that the intermediate flows can be recorded
        s = "FlowNet2/FlowNetCSS"
        c = refm.flownet_c(wts, inp, s + "/FlowNetCS/FlowNetC")["flow"]
        cs = refm.flownet_s(wts, refm._stack_inputs(inp, c), s + "/FlowNetCS/FlowNetS")["flow"]
        css = refm.flownet_s(wts, refm._stack_inputs(inp, cs), s + "/FlowNetS")["flow"]
        sd = refm.flownet_sd(wts, inp, "FlowNet2/FlowNetSD")["flow"]
        out = refm.flownet2(wts, inp)["flow"]
        ys, xs = probes(name)
        np.savez_compressed(os.path.join(HERE, "flownet2_%s_golden.npz" % name), probe_y=ys, probe_x=xs,
                            flow=out[0, ys, xs], flow_c=c[0, ys, xs], flow_cs=cs[0, ys, xs], flow_css=css[0, ys, xs],
                            flow_sd=sd[0, ys, xs], mean_mag=np.sqrt((out ** 2).sum(-1)).mean())
        print("wrote flownet2_%s_golden.npz in %.0f s; mean |flow| = %.4f px" %
              (name, time.time() - t0, np.sqrt((out ** 2).sum(-1)).mean()), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(SHAPES))
