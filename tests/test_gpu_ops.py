"""Parity of the three custom ops (through the reference's Python surface and the
C ABI) against the CPU oracle.  float32; tolerances stated per test."""
import numpy as np
import pytest
import torch

from oracle import ops as ref

pytestmark = pytest.mark.gpu


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


@pytest.fixture(scope="module")
def surf():
    from src.correlation import correlation
    from src.flow_warp import flow_warp
    from src.downsample import downsample
    return correlation, flow_warp, downsample


# ---- correlation ------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,args", [
    ((2, 12, 16, 32), (1, 20, 1, 2, 20)),     # call-site attrs (flownet_c.py:40) -> MFMA kernel, C=32
    ((2, 16, 16, 32), (1, 20, 1, 2, 20)),     # the same on the row-quad kernel (corr3)
    ((1, 6, 10, 256), (1, 20, 1, 2, 20)),     # call-site attrs, C=256, W < one tile
    ((1, 9, 70, 64), (1, 20, 1, 2, 20)),      # W > 64: two x blocks, ragged second block
    ((1, 8, 70, 64), (1, 20, 1, 2, 20)),      # H % 4 == 0 -> row-pair kernel (corr3), ragged second x block
    ((2, 4, 7, 32), (1, 20, 1, 2, 20)),       # corr3, image smaller than the displacement range in both directions
    ((1, 24, 130, 96), (1, 20, 1, 2, 20)),    # corr3, three x blocks, C = 96 (3 lines)
    ((2, 7, 9, 16), (1, 4, 1, 1, 4)),         # s2 = 1 band, fp32 MFMA with 1 slab
    ((1, 12, 14, 8), (3, 2, 2, 1, 3)),        # k=3, s1=2 -> generic kernel
    ((1, 8, 8, 5), (1, 3, 1, 3, 3)),          # odd channel count -> generic kernel
])
def test_correlation_matches_oracle(surf, shape, args):
    a, b = rnd(shape, 0), rnd(shape, 1)
    want = ref.correlation(a, b, *args)
    got = surf[0](a, b, *args)
    assert isinstance(got, np.ndarray) and got.shape == want.shape and got.dtype == np.float32
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6)  # fp32 summation-order tolerance
    gt = surf[0](torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), *args)
    assert gt.is_cuda and torch.equal(gt.cpu(), torch.from_numpy(got))


def test_correlation_at_the_flownet_c_call_site_size(surf):
    """The drop-in op at the REAL call-site size of FlowNetC at 512x384 (flownet_c.py:40: conv3 outputs N x 48 x 64 x 256
    -> N x 48 x 64 x 441), batch 2, against the oracle; plus the size-independent properties: bilinearity in a,
    and out[..., centre displacement] = mean_c a*b."""
    a, b = rnd((2, 48, 64, 256), 3), rnd((2, 48, 64, 256), 4)
    want = ref.correlation(a, b, 1, 20, 1, 2, 20)
    got = surf[0](a, b, 1, 20, 1, 2, 20)
    assert got.shape == (2, 48, 64, 441)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(got[..., 220], (a * b).mean(-1), rtol=1e-5, atol=2e-6)   # displacement (0, 0)
    got2 = surf[0]((2 * a).astype(np.float32), b, 1, 20, 1, 2, 20)
    np.testing.assert_allclose(got2, 2 * got, rtol=1e-6, atol=1e-7)  # linear in a (exact but for fp16-subnormal lo parts)


def test_correlation_validation(surf):
    a = rnd((1, 4, 4, 3), 0)
    with pytest.raises(ValueError):
        surf[0](a, a, 2, 2, 1, 1, 2)
    with pytest.raises(ValueError):
        surf[0](a[0], a[0], 1, 2, 1, 1, 2)
    with pytest.raises(ValueError):
        surf[0](a, a[:, :3], 1, 2, 1, 1, 2)
    with pytest.raises(ValueError):
        surf[0](a, a, 1, 8, 1, 1, 0)


@pytest.mark.parametrize("shape,args", [((1, 5, 6, 4), (1, 2, 1, 1, 2)), ((2, 6, 7, 3), (1, 4, 1, 2, 4)),
                                        ((1, 6, 7, 2), (3, 2, 2, 1, 3)),
                                        # the FlowNetC attribute set (flownet_c.py:40) -> the tiled kernels: ragged
                                        # 16-pixel tiles, rows displaced out of the image, C < 256 and C > 256
                                        ((2, 11, 37, 40), (1, 20, 1, 2, 20)), ((1, 5, 18, 300), (1, 20, 1, 2, 20)),
                                        ((1, 48, 64, 8), (1, 20, 1, 2, 20))])
def test_correlation_grad_matches_oracle(surf, shape, args):
    a, b = rnd(shape, 2), rnd(shape, 3)
    g = rnd(ref.correlation(a, b, *args).shape, 4)
    da, db = ref.correlation_grad(g, a, b, *args)
    ta = torch.from_numpy(a).cuda().requires_grad_(True)
    tb = torch.from_numpy(b).cuda().requires_grad_(True)
    out = surf[0](ta, tb, *args)
    out.backward(torch.from_numpy(g).cuda())
    atol = 2e-6 if args[1] < 20 else 1e-5  # 441-term fp32 sums with cancellation at the FlowNetC attributes
    np.testing.assert_allclose(ta.grad.cpu().numpy(), da, rtol=1e-5, atol=atol)
    np.testing.assert_allclose(tb.grad.cpu().numpy(), db, rtol=1e-5, atol=atol)


# ---- flow_warp ----------------------------------------------------------------------------------
def _warp_case(shape=(2, 9, 11, 3), seed=7):
    img = rnd(shape, seed)
    flow = rnd(shape[:3] + (2,), seed + 1, 3.0)
    flow[0, 0, 0] = (-0.5, 0.0)
    flow[0, 1, shape[2] - 1] = (0.5, 0.0)
    flow[0, shape[1] - 1, 5] = (0.0, 0.75)
    flow[0, 2, 2] = (1.0, -1.0)
    flow[0, 3, 3] = (np.nan, 0.0)
    flow[-1, 4, 4] = (100.0, 0.0)
    return img, flow


@pytest.mark.parametrize("shape", [(2, 9, 11, 3), (1, 33, 70, 3), (1, 8, 9, 5)])
def test_flow_warp_matches_oracle(surf, shape):
    img, flow = _warp_case(shape)
    want = ref.flow_warp(img, flow)
    got = surf[1](img, flow)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)  # FMA contraction only
    assert np.all(got[0, 0, 0] == 0) and np.all(got[0, 3, 3] == 0)


def test_flow_warp_grad_matches_oracle(surf):
    img, flow = _warp_case((2, 9, 11, 3))
    flow = np.nan_to_num(flow)
    g = rnd(img.shape, 11)
    dI, dF = ref.flow_warp_grad(img, flow, g)
    ti = torch.from_numpy(img).cuda().requires_grad_(True)
    tf_ = torch.from_numpy(flow).cuda().requires_grad_(True)
    surf[1](ti, tf_).backward(torch.from_numpy(g).cuda())
    np.testing.assert_allclose(ti.grad.cpu().numpy(), dI, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tf_.grad.cpu().numpy(), dF, rtol=1e-5, atol=1e-5)


def test_flow_warp_validation(surf):
    img, flow = _warp_case()
    with pytest.raises(ValueError):
        surf[1](img[0], flow)
    with pytest.raises(ValueError):
        surf[1](img, flow[:, :8])
    with pytest.raises(ValueError):
        surf[1](img, np.concatenate([flow, flow], -1))


# ---- downsample -----------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,size", [((2, 64, 96, 2), (4, 6)), ((1, 64, 96, 2), (16, 24)), ((3, 64, 96, 1), (4, 6)),
                                        ((1, 70, 100, 2), (5, 13)), ((2, 48, 64, 3), (3, 4)),
                                        ((1, 13, 17, 3), (5, 4)), ((1, 8, 8, 2), (8, 8)),
                                        ((1, 200, 40, 16), (3, 30))])  # last: window too large for LDS -> generic kernel
def test_downsample_matches_oracle(surf, shape, size):
    x = rnd(shape, 12, 4.0)
    x[0, 2:7, 3:9, 0] = np.nan
    want = ref.downsample(x, size)
    got = surf[2](x, size)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(want), rtol=2e-5, atol=2e-6)


def test_downsample_full_size_levels(surf):
    # BASELINE config sizes: 384x512 ground truth to the five FlowNetS loss scales (flownet_s.py:129-156)
    x = rnd((1, 384, 512, 2), 13, 5.0)
    x[0, 100:140, 200:260] = np.nan
    for size in ((6, 8), (12, 16), (24, 32), (48, 64), (96, 128)):
        want = ref.downsample(x, size)
        got = surf[2](x, size)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(want), rtol=5e-5, atol=5e-6)
    with pytest.raises(ValueError):
        surf[2](x, (2, 2, 2))
    with pytest.raises(ValueError):
        surf[2](x[0], (2, 2))
