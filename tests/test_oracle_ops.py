"""The two independent restatements of each custom op agree (oracle/ops.py):
the literal per-thread loop form vs the vectorised form; plus analytic
properties and the reference's argument validation."""
import numpy as np
import pytest

from oracle import ops


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


@pytest.mark.parametrize("cfg", [
    dict(shape=(2, 5, 6, 8), k=1, md=4, s1=1, s2=2, pad=4),   # call-site structure (flownet_c.py:40), small
    dict(shape=(1, 6, 7, 5), k=3, md=2, s1=2, s2=1, pad=3),   # kernel 3, stride_1 2
    dict(shape=(1, 4, 4, 40), k=1, md=3, s1=1, s2=3, pad=3),  # C > 32: strided partial sums
])
def test_correlation_vectorised_vs_loops(cfg):
    a, b = rnd(cfg["shape"], 0), rnd(cfg["shape"], 1)
    args = (cfg["k"], cfg["md"], cfg["s1"], cfg["s2"], cfg["pad"])
    v = ops.correlation(a, b, *args)
    l = ops.correlation_loops(a, b, *args)
    g = ops.correlation_geometry(cfg["shape"][1], cfg["shape"][2], *args)
    assert v.shape == l.shape == (cfg["shape"][0], g["oh"], g["ow"], g["D"])
    np.testing.assert_allclose(v, l, rtol=1e-5, atol=1e-6)


def test_correlation_callsite_shape_and_center():
    a, b = rnd((1, 12, 16, 16), 2), rnd((1, 12, 16, 16), 3)
    out = ops.correlation(a, b, 1, 20, 1, 2, 20)
    assert out.shape == (1, 12, 16, 441)
    # zero displacement is channel (10*21 + 10); x-displacement is the fast index
    np.testing.assert_allclose(out[..., 220], (a * b).mean(-1), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out[0, 3, 4, 221], (a[0, 3, 4] * b[0, 3, 6]).mean(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out[0, 3, 4, 220 + 21], (a[0, 3, 4] * b[0, 5, 4]).mean(), rtol=1e-5, atol=1e-6)
    assert out[0, 0, 0, 0] == 0  # displaced fully into the zero padding


def test_correlation_validation():
    a = rnd((1, 4, 4, 3), 0)
    with pytest.raises(ValueError):
        ops.correlation(a, a, 2, 2, 1, 1, 2)  # even kernel, correlation_kernel.cc:23
    with pytest.raises(ValueError):
        ops.correlation(a[0], a[0], 1, 2, 1, 1, 2)  # rank
    with pytest.raises(ValueError):
        ops.correlation(a, a[:, :3], 1, 2, 1, 1, 2)  # shape mismatch
    with pytest.raises(ValueError):
        ops.correlation(a, a, 1, 8, 1, 1, 0)  # output height < 1, :53


def test_correlation_grad_vectorised_vs_loops_and_numeric():
    a, b = rnd((1, 4, 5, 3), 4), rnd((1, 4, 5, 3), 5)
    args = (1, 2, 1, 1, 2)
    g = rnd(ops.correlation(a, b, *args).shape, 6)
    da_v, db_v = ops.correlation_grad(g, a, b, *args)
    da_l, db_l = ops.correlation_grad_loops(g, a, b, *args)
    np.testing.assert_allclose(da_v, da_l, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(db_v, db_l, rtol=1e-5, atol=1e-6)
    # k=1: the reference formula is the true gradient of sum(out*g)
    eps = 1e-2
    for (n, y, x, c) in [(0, 0, 0, 0), (0, 2, 3, 1), (0, 3, 4, 2)]:
        ap = a.copy(); ap[n, y, x, c] += eps
        am = a.copy(); am[n, y, x, c] -= eps
        num = ((ops.correlation(ap, b, *args).astype(np.float64) -
                ops.correlation(am, b, *args)) * g).sum() / (2 * eps)
        assert abs(num - da_v[n, y, x, c]) < 2e-3
        bp = b.copy(); bp[n, y, x, c] += eps
        bm = b.copy(); bm[n, y, x, c] -= eps
        num = ((ops.correlation(a, bp, *args).astype(np.float64) -
                ops.correlation(a, bm, *args)) * g).sum() / (2 * eps)
        assert abs(num - db_v[n, y, x, c]) < 2e-3


def _warp_case():
    img = rnd((2, 9, 11, 3), 7)
    flow = rnd((2, 9, 11, 2), 8, scale=3.0)
    flow[0, 0, 0] = (-0.5, 0.0)       # x2 in (-1, 0): exactly 0 in the reference
    flow[0, 1, 10] = (0.5, 0.0)       # x2 in (W-1, W): clamped neighbour, not zero
    flow[0, 8, 5] = (0.0, 0.75)       # y2 in (H-1, H)
    flow[0, 2, 2] = (1.0, -1.0)       # exact integers
    flow[0, 3, 3] = (np.nan, 0.0)     # NaN fails the range test
    flow[1, 4, 4] = (100.0, 0.0)      # far outside
    flow[1, 0, 0] = (0.0, 0.0)        # identity
    return img, flow


def test_flow_warp_vectorised_vs_loops_and_edges():
    img, flow = _warp_case()
    v = ops.flow_warp(img, flow)
    l = ops.flow_warp_loops(img, flow)
    np.testing.assert_allclose(v, l, rtol=1e-6, atol=1e-7)
    assert np.all(v[0, 0, 0] == 0) and np.all(v[0, 3, 3] == 0) and np.all(v[1, 4, 4] == 0)
    np.testing.assert_allclose(v[0, 1, 10], img[0, 1, 10], rtol=1e-6)   # replicated edge (A.2)
    np.testing.assert_allclose(v[0, 8, 5], img[0, 8, 5], rtol=1e-6)
    np.testing.assert_array_equal(v[0, 2, 2], img[0, 1, 3])
    np.testing.assert_array_equal(v[1, 0, 0], img[1, 0, 0])


def test_flow_warp_validation():
    img, flow = _warp_case()
    with pytest.raises(ValueError):
        ops.flow_warp(img[0], flow)
    with pytest.raises(ValueError):
        ops.flow_warp(img, flow[:, :8])
    with pytest.raises(ValueError):
        ops.flow_warp(img, np.concatenate([flow, flow], -1))


def test_flow_warp_grad_numeric():
    img = rnd((1, 6, 7, 2), 9)
    flow = (np.random.default_rng(10).uniform(-1.3, 1.3, (1, 6, 7, 2))).astype(np.float32)
    flow = np.where(np.abs(flow - np.round(flow)) < 0.1, flow + 0.25, flow).astype(np.float32)
    g = rnd((1, 6, 7, 2), 11)
    dI, dF = ops.flow_warp_grad(img, flow, g)
    eps = 1e-3

    def loss(i, f):
        return float((ops.flow_warp(i, f).astype(np.float64) * g).sum())

    for (y, x, c) in [(2, 3, 0), (0, 0, 1), (5, 6, 0)]:
        ip = img.copy(); ip[0, y, x, c] += eps
        im = img.copy(); im[0, y, x, c] -= eps
        assert abs((loss(ip, flow) - loss(im, flow)) / (2 * eps) - dI[0, y, x, c]) < 5e-3
    # interior pixels, away from the clamp (at the clamp the reference's gamma
    # formula is not the analytic derivative; it is restated as written)
    for (y, x) in [(2, 3), (3, 2)]:
        for c in (0, 1):
            fp = flow.copy(); fp[0, y, x, c] += eps
            fm = flow.copy(); fm[0, y, x, c] -= eps
            num = (loss(img, fp) - loss(img, fm)) / (2 * eps)
            assert abs(num - dF[0, y, x, c]) < 5e-3, (y, x, c, num, dF[0, y, x, c])


@pytest.mark.parametrize("shape,size", [((2, 16, 24, 2), (4, 6)), ((1, 13, 17, 3), (5, 4)),
                                        ((1, 8, 8, 2), (8, 8))])
def test_downsample_vectorised_vs_loops(shape, size):
    x = rnd(shape, 12, 4.0)
    x[0, 2:5, 3:6, 0] = np.nan   # NaN patch
    v = ops.downsample(x, size)
    l = ops.downsample_loops(x, size)
    assert v.shape == (shape[0], size[0], size[1], shape[3])
    assert np.array_equal(np.isnan(v), np.isnan(l))
    np.testing.assert_allclose(np.nan_to_num(v), np.nan_to_num(l), rtol=2e-5, atol=2e-6)


def test_downsample_nan_majority_and_identity():
    x = rnd((1, 8, 8, 2), 13)
    np.testing.assert_allclose(ops.downsample(x, (8, 8)), x, rtol=1e-6)  # FlowNet2 loss: identity scale
    x[0, :, :, 1] = np.nan
    out = ops.downsample(x, (2, 2))
    assert np.all(np.isnan(out[..., 1])) and not np.any(np.isnan(out[..., 0]))
    with pytest.raises(ValueError):
        ops.downsample(x, (2, 2, 2))
    with pytest.raises(ValueError):
        ops.downsample(x[0], (2, 2))
