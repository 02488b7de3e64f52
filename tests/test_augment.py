"""Training-input augmentation (the reference's preprocessing plugin, src/ops/preprocessing): oracle forms against
each other on the CPU, host coefficient logic, and the HIP passes against the oracle on the GPU."""
import numpy as np
import pytest

from oracle import augment as ref


def _case(seed, N=2, SH=20, SW=26, C=3, OH=12, OW=16):
    rng = np.random.default_rng(seed)
    src = rng.random((N, SH, SW, C), dtype=np.float32)
    coeffs = [dict(dx=0.05, dy=-0.03, angle=0.15, zoom_x=1.1, zoom_y=0.95), dict(angle=-0.4, zoom_x=0.7, zoom_y=0.7)][:N]
    trans = np.stack([ref.transmat_from_coeff(c, OW, OH, SW, SH) for c in coeffs])
    chroma = np.array([[1.2, 0.05, 1.3, 0.9, 1.1, 1.0], [0.8, -0.1, 0.7, 1.2, 0.8, 1.05]], np.float32)[:N]
    return src, trans, chroma, OH, OW


def test_oracle_forms_agree():
    src, trans, chroma, OH, OW = _case(0)
    for ch in (None, chroma):
        np.testing.assert_allclose(ref.augment(src, trans, ch, OH, OW), ref.augment_loops(src, trans, ch, OH, OW),
                                   rtol=2e-6, atol=2e-6)
    flows = (np.random.default_rng(1).standard_normal((2, 20, 26, 2)) * 3).astype(np.float32)
    inv_b = np.stack([ref.transmat_inverse(t) for t in trans[::-1]])
    np.testing.assert_allclose(ref.flow_augmentation(flows, trans, inv_b, OH, OW),
                               ref.flow_augmentation_loops(flows, trans, inv_b, OH, OW), rtol=1e-6, atol=1e-5)


def test_identity_and_inverse_properties():
    """Size-independent properties: identity coefficients with crop == source reproduce the image (up to the
    1.05 edge clamp) and leave the flow unchanged; T^-1 T = I; a pure zoom scales the flow."""
    rng = np.random.default_rng(2)
    src = rng.random((1, 10, 14, 3), dtype=np.float32)
    t = ref.transmat_from_coeff({}, 14, 10, 14, 10)
    np.testing.assert_allclose(t, [1, 0, 0, 0, 1, 0], atol=1e-6)
    out = ref.augment(src, [t], None, 10, 14)
    np.testing.assert_allclose(out[:, :-1, :-1], src[:, :-1, :-1], atol=1e-6)
    flows = (rng.standard_normal((1, 10, 14, 2))).astype(np.float32)
    np.testing.assert_allclose(ref.flow_augmentation(flows, [t], [ref.transmat_inverse(t)], 10, 14), flows, atol=1e-5)
    tz = ref.transmat_from_coeff(dict(dx=0.1, angle=0.3, zoom_x=1.3, zoom_y=0.8), 14, 10, 14, 10)
    i = ref.transmat_inverse(tz)
    m = np.array([[tz[0], tz[1], tz[2]], [tz[3], tz[4], tz[5]], [0, 0, 1]], np.float64)
    mi = np.array([[i[0], i[1], i[2]], [i[3], i[4], i[5]], [0, 0, 1]], np.float64)
    np.testing.assert_allclose(mi @ m, np.eye(3), atol=1e-4)
    zoom2 = ref.transmat_from_coeff(dict(zoom_x=2.0, zoom_y=2.0), 14, 10, 14, 10)
    const = np.full((1, 10, 14, 2), 1.0, np.float32)
    got = ref.flow_augmentation(const, [zoom2], [ref.transmat_inverse(zoom2)], 10, 14)
    np.testing.assert_allclose(got, 2.0, atol=1e-4)  # zooming in by 2 doubles the apparent motion


def test_host_coefficient_logic():
    from src import preprocessing as P
    rng = np.random.default_rng(0)
    par = dict(rand_type="uniform_bernoulli", exp=True, mean=0.2, spread=0.4, prob=1.0)
    vals = np.array([P.rng_generate(rng, par, 1.0, 1.0) for _ in range(2000)])
    assert np.exp(-0.2) <= vals.min() and vals.max() <= np.exp(0.6)
    assert P.rng_generate(rng, dict(par, prob=0.0), 1.0, 7.0) == 7.0
    assert P.rng_generate(rng, dict(par, spread=0.0, exp=False), 1.0, 0.0) == pytest.approx(0.2)
    g = np.array([P.rng_generate(rng, dict(par, rand_type="gaussian_bernoulli", exp=False, mean=1.0, spread=0.5), 0.5, 0.0)
                  for _ in range(4000)])
    assert abs(g.mean() - 1.0) < 0.02 and abs(g.std() - 0.25) < 0.02       # spread * discount
    with pytest.raises(ValueError):
        P.rng_generate(rng, dict(par, rand_type="cauchy"), 1.0, 0.0)
    c = dict(dx=0.05, dy=-0.03, angle=0.15, zoom_x=1.1, zoom_y=0.95)
    np.testing.assert_array_equal(P.transmat_from_coeff(c, 448, 384, 512, 384), ref.transmat_from_coeff(c, 448, 384, 512, 384))
    assert P.corners_fit({}, 512, 384, 448, 320) == ref.corners_fit({}, 512, 384, 448, 320) is True
    # crop height == source height: the bottom corners land on row 383 > 384 - 2, so the identity is rejected
    assert P.corners_fit({}, 512, 384, 448, 384) == ref.corners_fit({}, 512, 384, 448, 384) is False
    big = dict(dx=0.4)
    assert P.corners_fit(big, 512, 384, 448, 320) == ref.corners_fit(big, 512, 384, 448, 320) is False
    # combine_with multiplies: image b's draw times image a's value; a cleared brightness reads as 0
    assert P._combine(dict(dx=0.5), dict(dx=0.2, brightness=0.3, gamma=1.5)) == dict(dx=0.1, brightness=0.0, gamma=1.5)
    aug = P._params(["translate", "zoom", "noise"], ["uniform_bernoulli"] * 3, [False, True, False], [0, 0.2, 0.03],
                    [0.4, 0.4, 0.03], [1.0, 1.0, 1.0])
    assert set(aug) == {"translate", "zoom"}
    for _ in range(20):
        cc = P.generate_valid_spatial_coeffs(rng, aug, 1.0, {}, 512, 384, 448, 384)
        assert P.corners_fit(cc, 512, 384, 448, 384) and cc["zoom_x"] == cc["zoom_y"]
    assert P._discount((1000.0, 0.5, 1.0), 0) == pytest.approx(0.5)
    assert P._discount((1000.0, 0.5, 1.0), 10 ** 7) == pytest.approx(1.0)


@pytest.mark.gpu
def test_hip_augment_matches_oracle():
    from src import preprocessing as P
    src, trans, chroma, OH, OW = _case(3, SH=40, SW=52, OH=24, OW=32)
    for ch in (None, chroma):
        got = P.augment(src, trans, ch, (OH, OW)).cpu().numpy()
        np.testing.assert_allclose(got, ref.augment(src, trans, ch, OH, OW), rtol=1e-5, atol=1e-5)
    gray = src[..., :1].copy()                                     # generic channel count, no chromatic part
    np.testing.assert_allclose(P.augment(gray, trans, None, (OH, OW)).cpu().numpy(),
                               ref.augment(gray, trans, None, OH, OW), rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        P.augment(gray, trans, chroma, (OH, OW))
    flows = (np.random.default_rng(4).standard_normal((2, 40, 52, 2)) * 3).astype(np.float32)
    inv_b = np.stack([ref.transmat_inverse(t) for t in trans[::-1]])
    got = P.flow_augmentation(flows, trans, inv_b, (OH, OW)).cpu().numpy()
    np.testing.assert_allclose(got, ref.flow_augmentation(flows, trans, inv_b, OH, OW), rtol=1e-5, atol=1e-4)
    with pytest.raises(ValueError):
        P.flow_augmentation(flows, trans[:1], inv_b, (OH, OW))


@pytest.mark.gpu
def test_data_augmentation_op_end_to_end():
    """The op surface with FlyingChairs-like parameters: outputs have the crop size, image b's transform composes
    with image a's, and warping the identity flow through (T_a, T_b^-1) gives the flow between the two crops."""
    from src import preprocessing as P
    rng = np.random.default_rng(5)
    a = rng.random((4, 96, 128, 3), dtype=np.float32)
    names_a = ["translate", "rotate", "zoom", "squeeze"]
    names_b = ["translate", "rotate", "zoom", "gamma", "brightness", "contrast", "color"]
    args_a = (names_a, ["uniform_bernoulli"] * 4, [False, False, True, True], [0, 0, 0.2, 0], [0.2, 0.2, 0.3, 0.2], [1.0] * 4, [])
    args_b = (names_b, ["gaussian_bernoulli"] * 7, [False, False, True, True, False, True, True], [0] * 7,
              [0.03, 0.03, 0.03, 0.02, 0.02, 0.02, 0.02], [1.0] * 7, [])
    oa, ob, ta, itb = P.data_augmentation(a, a, 0, (64, 96), *args_a, *args_b, seed=123)
    assert oa.shape == ob.shape == (4, 64, 96, 3) and ta.shape == itb.shape == (4, 6)
    assert float(oa.min()) >= 0.0 and float(ob.max()) <= 1.0
    np.testing.assert_allclose(oa.cpu().numpy(), ref.augment(a, ta.numpy(), None, 64, 96), rtol=1e-5, atol=1e-5)
    zero = np.zeros((4, 96, 128, 2), np.float32)
    f = P.flow_augmentation(zero, ta, itb, (64, 96)).cpu().numpy()
    np.testing.assert_allclose(f, ref.flow_augmentation(zero, ta.numpy(), itb.numpy(), 64, 96), rtol=1e-5, atol=1e-4)
    # (no bound on |f|: combine_with MULTIPLIES image a's coefficients into image b's draw, so b's translation is
    # ~0.03 * dx_a rather than dx_a + 0.03 -- the induced flow between the crops is large; reference behaviour)
    oa2, _, ta2, _ = P.data_augmentation(a, a, 0, (64, 96), *args_a, *args_b, seed=123)
    assert torch_equal(oa, oa2) and np.array_equal(ta.numpy(), ta2.numpy())   # seeded: reproducible


def torch_equal(x, y):
    import torch
    return bool(torch.equal(x, y))
