"""MPI-Sintel metric code of the host library against outputs of the reference's own flowlib
(tests/golden/make_golden_metrics.py ran the reference; nothing here imports it)."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "metrics_golden.npz"))


def _check(g, tag, est, gt, occ, inv):
    from src import flowlib
    m, not_occ, z0, z1, z2 = flowlib.compute_all_metrics(est.copy(), gt.copy(), occ_mask=occ, inv_mask=inv)
    keys = [str(k) for k in g[tag + "_keys"]]
    got = np.array([float(m[k]) for k in keys])
    np.testing.assert_allclose(got, g[tag + "_values"], rtol=2e-6, atol=1e-7)
    assert [not_occ, z0, z1, z2] == list(g[tag + "_counts"])
    assert flowlib.get_metrics(m, flow_fname="frame_0001") == str(g[tag + "_text"])
    return m


def test_compute_all_metrics_matches_reference_outputs(g):
    from src import flowlib
    gt, est, occ, inv = g["gt"], g["est"], g["occ"], g["inv"]
    m = _check(g, "full", est, gt, occ, inv)
    _check(g, "nomask", est, gt, None, None)
    _check(g, "small", est, np.clip(gt, -3, 3), occ, None)
    assert flowlib.get_metrics(m, average=True) == str(g["avg_text"])


def test_flow_error_blocks_match_reference_outputs(g):
    from src import flowlib
    gt, est, mask = g["gt"], g["est"], g["occ"] == 255
    args = lambda: (gt[..., 0].copy(), gt[..., 1].copy(), est[..., 0].copy(), est[..., 1].copy())
    np.testing.assert_allclose(flowlib.flow_error_mask(*args(), mask, True), g["fem_ignore_true"], rtol=2e-6)
    np.testing.assert_allclose(flowlib.flow_error_mask(*args(), mask, False), g["fem_ignore_false"], rtol=2e-6)
    got = np.array(flowlib.flow_error(*args()), np.float64)
    np.testing.assert_allclose(got, g["flow_error"], rtol=2e-6, equal_nan=True)  # NaN angles reproduced
    assert flowlib.evaluate_flow(gt, est) == pytest.approx(float(g["flow_error"][2]), rel=2e-6)


def test_sequence_average_follows_the_reference_formulas():
    """net.py:958-984, quirks included (see Net._average_metrics)."""
    from src.net import Net
    t = np.arange(24, dtype=np.float64).reshape(2, 12) + 1.0
    t[1, 9] = np.nan
    avg = Net._average_metrics(t.copy(), np.array([1, 1, 1, 1]))
    assert avg[0] == pytest.approx((1 + 13) / 2)
    np.testing.assert_allclose(avg[6:9], 0.0)                        # (1 - #frames without occlusions) = 0
    assert avg[9] == pytest.approx(10.0 / 3 * (3 / (3 - 1)))         # divisor 2 non-inf + 1 NaN = 3, rescaled
    assert avg[10] == pytest.approx((11 + 23) / 2) and avg[11] == pytest.approx((12 + 24) / 2)  # never rescaled
    avg2 = Net._average_metrics(t.copy(), np.array([0, 0, 0, 0]))
    assert avg2[7] == pytest.approx((8 + 20) / 2)
