"""Generate tests/golden/plan_*.npz: oracle answers for the EXACT workloads bench.py times, at their own batch sizes
(BASELINE configs 2-5), on bench.py's own synthetic inputs (bench.synth_pairs / synth_gt, rank 0) and seeded weights
(src.weights.init_weights(model, 1234)).

  plan_flownetc_b8_384x512        FlowNetC forward, 8 pairs (flownet_c.py:15-125): 4096 probe pixels of `flow` per pair +
                                  predict_flow6 in full
  plan_flownets_b8_384x512        FlowNetS forward, 8 pairs (flownet_s.py:14-120): the same
  plan_flownet2_b4_384x512        FlowNet2 full stack, 4 pairs (flownet2.py:18-105): probes of the final flow and of the
                                  C / CS / CSS / SD flows (a deviation is localised to a sub-network)
  plan_flownet2_b4_448x1024       the same at the Sintel shape: 4 pairs of 436 x 1024 zero-padded to 448 x 1024 as
                                  Net.adapt_x does (net.py:373-388)
  plan_flownets_train_b8_384x512  FlowNetS loss + all-layer gradients of the batch-8 train step (flownet_s.py:122-161 under
                                  tf.gradients): torch float64 autograd of the restated graph per pair, summed -- the loss is
                                  a sum over the batch divided by N (utils.py:214-224), so the batch gradient is the mean
                                  of the per-pair gradients.  Stored: the loss, every bias gradient in full, 2048 probe
                                  entries of every filter gradient (reference HWIO / HW-O-I layout) and its max |g|

The oracle is this build's restatement (parity unpinned): these files are the device-free target
the -m gpu tests of the timed plans compare with, not a pin against the reference.

    python tests/golden/make_golden_bench_plans.py [name ...]        (~6 minutes for all five on 8 cores)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-tf_amd")]

SEED_W = 1234
NPROBE = 4096
NGPROBE = 2048


def probes(h, w):
    rng = np.random.default_rng(4242)
    return rng.integers(0, h, NPROBE), rng.integers(0, w, NPROBE)


def forward_plan(model, batch, h, w, rows=None):
    import bench
    from oracle import models as refm
    from src import weights as W
    wts = W.init_weights(model, SEED_W)
    a, b = bench.synth_pairs(batch, rows or h, w, seed0=0)
    a, b = bench.pad64(a), bench.pad64(b)
    assert a.shape[1:3] == (h, w)
    ys, xs = probes(h, w)
    out = {"probe_y": ys, "probe_x": xs}
    keep = {}
    for i in range(batch):
        inp = {"input_a": a[i:i + 1], "input_b": b[i:i + 1]}
        if model == "FlowNet2":
            # flownet2.py:22-23 unrolled one level so that the intermediate flows can be recorded
            s = "FlowNet2/FlowNetCSS"
            c = refm.flownet_c(wts, inp, s + "/FlowNetCS/FlowNetC")["flow"]
            cs = refm.flownet_s(wts, refm._stack_inputs(inp, c), s + "/FlowNetCS/FlowNetS")["flow"]
            css = refm.flownet_s(wts, refm._stack_inputs(inp, cs), s + "/FlowNetS")["flow"]
            sd = refm.flownet_sd(wts, inp, "FlowNet2/FlowNetSD")["flow"]
            res = {"flow": refm.flownet2(wts, inp)["flow"], "flow_c": c, "flow_cs": cs, "flow_css": css, "flow_sd": sd}
        else:
            r = refm.MODELS[model](wts, inp)
            res = {"flow": r["flow"]}
            keep.setdefault("predict_flow6", []).append(r["predict_flow6"][0].astype(np.float32))
        for k, v in res.items():
            keep.setdefault(k, []).append(v[0, ys, xs].astype(np.float32))
        keep.setdefault("mean_mag", []).append(np.sqrt((res["flow"] ** 2).sum(-1)).mean())
        print("  %s pair %d: mean |flow| = %.4f px" % (model, i, keep["mean_mag"][-1]), flush=True)
    out.update({k: np.stack(v) for k, v in keep.items()})
    return out


def train_plan(batch, h, w):
    import bench
    from oracle import train as reft
    from src import weights as W
    wts = W.init_weights("FlowNetS", SEED_W)
    a, b = bench.synth_pairs(batch, h, w, seed0=0)
    gt = bench.synth_gt(batch, h, w, 0)
    loss, grads = 0.0, None
    for i in range(batch):
        l, g, _ = reft.flownet_s_loss_and_grads(wts, a[i:i + 1], b[i:i + 1], gt[i:i + 1])
        loss += l / batch
        if grads is None:
            grads = {k: v / batch for k, v in g.items()}
        else:
            for k, v in g.items():
                grads[k] += v / batch
        print("  FlowNetS pair %d: loss %.6f" % (i, l), flush=True)
    out = {"loss": np.float64(loss)}
    rng = np.random.default_rng(99)
    for k in sorted(grads):
        g = grads[k]
        if k.endswith("/biases") or g.size <= NGPROBE:
            out[k] = g
        else:
            idx = np.sort(rng.choice(g.size, NGPROBE, replace=False))
            # the largest entries are what the tolerance is relative to: always probe the top 64 as well
            top = np.argsort(np.abs(g.reshape(-1)))[-64:]
            idx = np.unique(np.concatenate([idx, top]))
            out[k + "#idx"] = idx.astype(np.int64)
            out[k + "#val"] = g.reshape(-1)[idx]
            out[k + "#max"] = np.float64(np.abs(g).max())
            out[k + "#shape"] = np.asarray(g.shape, np.int64)
    return out


PLANS = {
    "plan_flownetc_b8_384x512": lambda: forward_plan("FlowNetC", 8, 384, 512),
    "plan_flownets_b8_384x512": lambda: forward_plan("FlowNetS", 8, 384, 512),
    "plan_flownet2_b4_384x512": lambda: forward_plan("FlowNet2", 4, 384, 512),
    "plan_flownet2_b4_448x1024": lambda: forward_plan("FlowNet2", 4, 448, 1024, rows=436),
    "plan_flownets_train_b8_384x512": lambda: train_plan(8, 384, 512),
}


def main(names):
    for name in names:
        t0 = time.time()
        out = PLANS[name]()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote %s.npz in %.0f s" % (name, time.time() - t0), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(PLANS))
