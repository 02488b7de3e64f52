"""Op-level golden vectors (tests/golden/ops_golden.npz): a regression pin of the oracle on CPU
-- the stored outputs come from the literal per-thread-loop restatements, the checks here run the vectorised ones --
and a device-free target for the HIP ops through the reference's Python op surface on the GPU."""
import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-tf_amd")]

_spec = importlib.util.spec_from_file_location("make_golden_ops", os.path.join(ROOT, "tests", "golden", "make_golden_ops.py"))
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)


@pytest.fixture(scope="module")
def gold():
    with np.load(os.path.join(ROOT, "tests", "golden", "ops_golden.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def cases(gold):
    c = G.cases()
    np.testing.assert_array_equal(c["warp_flow"], gold["warp_flow_input"])  # the hand-set flows did not drift
    return c


def _close(got, want, rtol=1e-5, atol=1e-6):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(want), rtol=rtol, atol=atol)


def test_oracle_vectorised_forms_reproduce_the_golden_vectors(gold, cases):
    from oracle import nn, ops
    c = cases
    _close(ops.correlation(c["corr_a1"], c["corr_b1"], *G.CALL_SITE), gold["corr1"])        # stored: the loop form
    _close(ops.correlation(c["corr_a3"], c["corr_b3"], 3, 2, 2, 1, 3), gold["corr3"])
    _close(ops.correlation(c["corr_a2"], c["corr_b2"], *G.CALL_SITE), gold["corr2"])
    da, db = ops.correlation_grad(c["corr_g1"], c["corr_a1"], c["corr_b1"], *G.CALL_SITE)
    _close(da, gold["corr1_da"]); _close(db, gold["corr1_db"])
    _close(ops.flow_warp(c["warp_img"], c["warp_flow"]), gold["warp"])                       # stored: the loop form
    di, df = ops.flow_warp_grad(c["warp_img"], c["warp_flow"], c["warp_grad"])
    _close(di, gold["warp_dimg"]); _close(df, gold["warp_dflow"])
    _close(ops.downsample(c["ds_in"], (6, 8)), gold["ds_6x8"])
    _close(ops.downsample(c["ds_in"], (96, 128)), gold["ds_96x128"])
    _close(nn.conv2d(c["conv_x"], c["conv_w"], c["conv_b"], stride=2, padding=3, activation=nn.leaky_relu), gold["conv7s2"])
    _close(nn.conv2d_transpose(c["deconv_x"], c["deconv_w"], activation=nn.leaky_relu), gold["deconv"])
    r = nn.resize_bilinear_align_corners(c["resize_x"] * 20.0, (384, 512))
    _close(r[0, gold["resize_probe_y"], gold["resize_probe_x"]], gold["resize_probes"])
    # the hand-set flows do what they were set for
    w = gold["warp"]
    assert np.all(w[0, 0, 0] == 0) and np.all(w[1, 2, 3] == 0) and np.all(w[1, 3, 3] == 0)  # outside / NaN -> 0
    np.testing.assert_allclose(w[0, 4, 4], c["warp_img"][0, 1, 6], rtol=1e-6)                # exact integer flow


@pytest.mark.gpu
def test_hip_ops_reproduce_the_golden_vectors(gold, cases):
    import torch
    from src.correlation import correlation
    from src.downsample import downsample
    from src.flow_warp import flow_warp
    c = cases
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    host = lambda t: t.detach().cpu().numpy()
    for a, b, args, key in (("corr_a1", "corr_b1", G.CALL_SITE, "corr1"), ("corr_a2", "corr_b2", G.CALL_SITE, "corr2"),
                            ("corr_a3", "corr_b3", (3, 2, 2, 1, 3), "corr3")):
        _close(host(correlation(dev(c[a]), dev(c[b]), *args)), gold[key], rtol=2e-5, atol=2e-6)
    ta, tb = dev(c["corr_a1"]).requires_grad_(True), dev(c["corr_b1"]).requires_grad_(True)
    correlation(ta, tb, *G.CALL_SITE).backward(dev(c["corr_g1"]))
    _close(host(ta.grad), gold["corr1_da"], rtol=1e-5, atol=1e-5)
    _close(host(tb.grad), gold["corr1_db"], rtol=1e-5, atol=1e-5)
    ti, tf = dev(c["warp_img"]).requires_grad_(True), dev(c["warp_flow"]).requires_grad_(True)
    out = flow_warp(ti, tf)
    _close(host(out), gold["warp"], rtol=1e-6, atol=1e-6)
    out.backward(dev(c["warp_grad"]))
    _close(host(ti.grad), gold["warp_dimg"], rtol=1e-5, atol=1e-5)
    _close(host(tf.grad), gold["warp_dflow"], rtol=1e-5, atol=1e-5)
    _close(host(downsample(dev(c["ds_in"]), [6, 8])), gold["ds_6x8"], rtol=5e-5, atol=5e-6)
    _close(host(downsample(dev(c["ds_in"]), [96, 128])), gold["ds_96x128"], rtol=5e-5, atol=5e-6)
