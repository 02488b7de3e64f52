"""Generate tests/golden/ops_golden.npz: the op-level vectors SURVEY.md section 8c lists -- correlation forward /
backward (12x16x8 and 6x10x256 at the FlowNetC call-site attributes (1, 20, 1, 2, 20), one (k=3, stride_1=2) case),
flow_warp forward / backward on 2x9x11x3 with flows that hit x2 < 0, x2 in [W-1, W), exact integers and NaN,
downsample 2x384x512x2 -> 6x8 and -> 96x128 with a NaN patch, one 7x7 stride-2 conv, one 4x4 stride-2 transposed conv
and one align_corners resize 96x128 -> 384x512.  Inputs are regenerated from the seeds below (only outputs and the few
hand-set inputs are stored), outputs come from the literal per-thread-loop restatements where they exist
(oracle/ops.py *_loops) and from the vectorised ones elsewhere.

The reference's kernels cannot run here (GPU-only TensorFlow plugins), so this file is a REGRESSION pin of the
oracle and a device-free target for the GPU tests -- not a pin against the reference.

    python tests/golden/make_golden_ops.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-tf_amd")]

CALL_SITE = (1, 20, 1, 2, 20)  # flownet_c.py:40


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def cases():
    """Every input of the fixture, from seeds (shared by the generator and the tests)."""
    c = {}
    c["corr_a1"], c["corr_b1"] = rnd((2, 12, 16, 8), 0), rnd((2, 12, 16, 8), 1)
    c["corr_a2"], c["corr_b2"] = rnd((2, 6, 10, 256), 2), rnd((2, 6, 10, 256), 3)
    c["corr_a3"], c["corr_b3"] = rnd((1, 9, 11, 5), 4), rnd((1, 9, 11, 5), 5)      # kernel_size 3, stride_1 2
    c["corr_g1"] = rnd((2, 12, 16, 441), 6)
    c["warp_img"] = rnd((2, 9, 11, 3), 7)
    flow = rnd((2, 9, 11, 2), 8, 3.0)
    flow[0, 0, 0] = (-0.5, 0.0)            # x2 < 0: outside
    flow[0, 1, 10] = (0.5, 0.0)            # x2 in [W-1, W): right tap clamped
    flow[0, 8, 5] = (0.0, 0.75)            # y2 in [H-1, H)
    flow[0, 4, 4] = (2.0, -3.0)            # exact integers
    flow[1, 2, 3] = (np.nan, 1.0)          # NaN fails the range test
    flow[1, 3, 3] = (11.0, 0.0)            # x2 == W: outside
    c["warp_flow"] = flow
    c["warp_grad"] = rnd((2, 9, 11, 3), 9)
    ds = rnd((2, 384, 512, 2), 10, 5.0)
    ds[0, 100:140, 200:260] = np.nan
    ds[1, 0:3, 0:3, 1] = np.nan
    c["ds_in"] = ds
    c["conv_x"], c["conv_w"], c["conv_b"] = rnd((1, 20, 28, 6), 11), rnd((7, 7, 6, 16), 12, 0.1), rnd((16,), 13, 0.1)
    c["deconv_x"], c["deconv_w"] = rnd((1, 6, 8, 12), 14), rnd((4, 4, 5, 12), 15, 0.2)
    c["resize_x"] = rnd((1, 96, 128, 2), 16)
    return c


def compute(c):
    from oracle import nn, ops
    out = {}
    out["corr1"] = ops.correlation_loops(c["corr_a1"], c["corr_b1"], *CALL_SITE)
    out["corr2"] = ops.correlation(c["corr_a2"], c["corr_b2"], *CALL_SITE)
    out["corr3"] = ops.correlation_loops(c["corr_a3"], c["corr_b3"], 3, 2, 2, 1, 3)
    out["corr1_da"], out["corr1_db"] = ops.correlation_grad(c["corr_g1"], c["corr_a1"], c["corr_b1"], *CALL_SITE)
    out["warp"] = ops.flow_warp_loops(c["warp_img"], c["warp_flow"])
    out["warp_dimg"], out["warp_dflow"] = ops.flow_warp_grad(c["warp_img"], c["warp_flow"], c["warp_grad"])
    out["ds_6x8"] = ops.downsample(c["ds_in"], (6, 8))
    out["ds_96x128"] = ops.downsample(c["ds_in"], (96, 128))
    out["conv7s2"] = nn.conv2d(c["conv_x"], c["conv_w"], c["conv_b"], stride=2, padding=3, activation=nn.leaky_relu)
    out["deconv"] = nn.conv2d_transpose(c["deconv_x"], c["deconv_w"], activation=nn.leaky_relu)
    out["resize"] = nn.resize_bilinear_align_corners(c["resize_x"] * 20.0, (384, 512))
    return out


def main():
    c = cases()
    out = compute(c)
    stored = {k: np.asarray(v) for k, v in out.items()}
    stored["warp_flow_input"] = c["warp_flow"]  # the hand-set flows, kept verbatim
    # the big results as float32 (what the ops produce); everything is < 1 MB compressed except resize: keep probes
    rng = np.random.default_rng(99)
    ys, xs = rng.integers(0, 384, 256), rng.integers(0, 512, 256)
    stored["resize_probes"] = stored.pop("resize")[0, ys, xs]
    stored["resize_probe_y"], stored["resize_probe_x"] = ys, xs
    np.savez_compressed(os.path.join(HERE, "ops_golden.npz"), **stored)
    print("wrote ops_golden.npz:", {k: v.shape for k, v in stored.items()})


if __name__ == "__main__":
    main()
