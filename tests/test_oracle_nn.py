"""Cross-check oracle/nn.py (restated TF op definitions) against
torch's CPU kernels as an independent implementation."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nn


def rnd(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape)


def nhwc(t):
    return t.permute(0, 2, 3, 1).numpy()


def nchw(a):
    return torch.from_numpy(np.ascontiguousarray(a)).permute(0, 3, 1, 2)


@pytest.mark.parametrize("k,s,p,cin,cout", [(7, 2, 3, 6, 8), (5, 2, 2, 8, 12), (3, 1, 1, 5, 7),
                                            (3, 2, 1, 9, 4), (1, 1, 0, 16, 3)])
def test_conv2d_matches_torch(k, s, p, cin, cout):
    x, w, b = rnd((2, 16, 24, cin), 0), rnd((k, k, cin, cout), 1), rnd((cout,), 2)
    ref = F.conv2d(nchw(x), torch.from_numpy(w).permute(3, 2, 0, 1), torch.from_numpy(b), stride=s, padding=p)
    out = nn.conv2d(x, w, b, stride=s, padding=p)
    np.testing.assert_allclose(out, nhwc(ref), rtol=1e-10, atol=1e-10)
    out = nn.conv2d(x, w, b, stride=s, padding=p, activation=nn.leaky_relu)
    np.testing.assert_allclose(out, nhwc(F.leaky_relu(ref, 0.1)), rtol=1e-10, atol=1e-10)


def test_conv2d_transpose_matches_torch():
    # slim.conv2d_transpose(k=4, s=2, VALID) + antipad(1) == ConvTranspose2d(k=4, s=2, padding=1, bias=False)
    # with weight[i,o,ky,kx] = Wt[ky,kx,o,i]   
    x, w = rnd((2, 5, 6, 7), 3), rnd((4, 4, 3, 7), 4)
    ref = F.conv_transpose2d(nchw(x), torch.from_numpy(w).permute(3, 2, 0, 1), stride=2, padding=1)
    out = nn.conv2d_transpose(x, w)
    assert out.shape == (2, 10, 12, 3)
    np.testing.assert_allclose(out, nhwc(ref), rtol=1e-10, atol=1e-10)


def test_resize_bilinear_align_corners_matches_torch():
    x = rnd((2, 6, 8, 2), 5)
    ref = F.interpolate(nchw(x), size=(24, 32), mode="bilinear", align_corners=True)
    np.testing.assert_allclose(nn.resize_bilinear_align_corners(x, (24, 32)), nhwc(ref), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(nn.resize_bilinear_align_corners(x, (6, 8)), x, rtol=0, atol=0)


def test_pad_antipad_leaky_channelnorm():
    x = rnd((1, 3, 4, 2), 6)
    assert nn.pad(x, 2).shape == (1, 7, 8, 2) and np.all(nn.pad(x, 2)[:, :2] == 0)
    np.testing.assert_array_equal(nn.antipad(nn.pad(x, 2), 2), x)
    np.testing.assert_allclose(nn.leaky_relu(x), np.maximum(x, 0.1 * x), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(nn.channel_norm(x)[..., 0], np.linalg.norm(x, axis=3), rtol=1e-12)


def test_config1_flownet_s_on_the_sample_pair_matches_its_fixture(golden_dir):
    """BASELINE config 1: FlowNetS forward on data/samples/0img0.ppm + 0img1.ppm at 512x384 through the NumPy
    oracle (seeded weights).  Regression pin of the oracle (tests/golden/make_golden_flownets.py)."""
    import os
    import sys
    sys.path.insert(0, golden_dir)
    import make_golden_flownets as gen
    from oracle import models as refm
    from src import weights as W
    g = np.load(os.path.join(golden_dir, "flownets_sample0_golden.npz"))
    a, b = gen.inputs()
    assert a.shape == (1, 384, 512, 3)
    out = refm.flownet_s(W.init_weights("FlowNetS", 1234), {"input_a": a, "input_b": b})
    np.testing.assert_allclose(out["predict_flow6"], g["predict_flow6"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(out["flow"][0, g["probe_y"], g["probe_x"]], g["flow_probes"], rtol=1e-9, atol=1e-12)
