"""Generate tests/golden/flownets_sample0_golden.npz: BASELINE config 1 -- FlowNetS forward on the reference's
sample pair (data/samples/0img0.ppm + 0img1.ppm, 512x384) through the NumPy oracle with seeded weights
(src.weights.init_weights('FlowNetS', 1234)).  Stored: predict_flow6 in full and 64 probe pixels of `flow`
(SURVEY.md section 8c).  The oracle is this build's restatement, so this fixture is a REGRESSION pin of the oracle
(and a device-free target for the GPU test), not a pin against the reference.

    python tests/golden/make_golden_flownets.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-tf_amd")]


def inputs():
    from PIL import Image
    a = np.asarray(Image.open(os.path.join(HERE, "samples", "0img0.ppm")).convert("RGB"), np.float32) / 255.0
    b = np.asarray(Image.open(os.path.join(HERE, "samples", "0img1.ppm")).convert("RGB"), np.float32) / 255.0
    return a[None], b[None]


def probes():
    rng = np.random.default_rng(42)
    return rng.integers(0, 384, 64), rng.integers(0, 512, 64)


def main():
    from oracle import models as refm
    from src import weights as W
    a, b = inputs()
    out = refm.flownet_s(W.init_weights("FlowNetS", 1234), {"input_a": a, "input_b": b})
    ys, xs = probes()
    np.savez_compressed(os.path.join(HERE, "flownets_sample0_golden.npz"),
                        predict_flow6=out["predict_flow6"].astype(np.float64),
                        flow_probes=out["flow"][0, ys, xs].astype(np.float64), probe_y=ys, probe_x=xs)
    print("wrote flownets_sample0_golden.npz; mean |flow| = %.4f" % np.abs(out["flow"]).mean())


if __name__ == "__main__":
    main()
