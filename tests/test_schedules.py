"""Learning-rate schedules / policies (reference src/training_schedules.py, src/utils.py:24-135): values at hand-computed
points of the formulas."""
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-tf_amd"))

from src import training_schedules as TS  # noqa: E402


def test_piecewise_schedules():
    lr = lambda s, t: TS.learning_rate(s, t)
    assert lr(TS.LONG_SCHEDULE, 0) == 1e-4 and lr(TS.LONG_SCHEDULE, 400000) == 1e-4   # x <= boundary keeps the rate
    assert lr(TS.LONG_SCHEDULE, 400001) == 5e-5 and lr(TS.LONG_SCHEDULE, 1100000) == 6.25e-6
    assert lr(TS.FINE_SCHEDULE, 250000) == 5e-6 and TS.FINE_SCHEDULE["max_iters"] == 500000
    assert lr(TS.SHORT_SCHEDULE, 450000) == 2.5e-5
    s3 = TS.SCHEDULES["finetune_sintel_s3"]      # training_schedules.py:110-118
    assert s3["step_values"][0] == 345000 and s3["step_values"][-1] == 440000 and s3["max_iters"] == 450000
    assert s3["learning_rates"][0] == 2e-5 and s3["learning_rates"][-1] == pytest.approx(1.953125e-08)
    k2 = TS.SCHEDULES["finetune_kitti_s2"]       # :141-149
    assert k2["step_values"][4] == 247500 and k2["learning_rates"][1] == 2e-5
    assert all(len(v["learning_rates"]) == len(v["step_values"]) + 1 for v in TS.SCHEDULES.values()
               if not isinstance(v["learning_rates"], str))


def test_cyclic_policies():
    f = TS.cyclic_lr
    # triangular: base at 0, max at step_size, base again at 2 step_size
    assert f(0, 1e-5, 1e-4, 1000, mode="triangular") == pytest.approx(1e-5)
    assert f(1000, 1e-5, 1e-4, 1000, mode="triangular") == pytest.approx(1e-4)
    assert f(500, 1e-5, 1e-4, 1000, mode="triangular") == pytest.approx(5.5e-5)
    assert f(2000, 1e-5, 1e-4, 1000, mode="triangular") == pytest.approx(1e-5)
    # triangular2: the second cycle's amplitude is halved
    assert f(3000, 1e-5, 1e-4, 1000, mode="triangular2") == pytest.approx(1e-5 + 0.5 * 9e-5)
    # exponential: amplitude * gamma ** step
    assert f(1000, 1e-5, 1e-4, 1000, gamma=0.999, mode="exponential") == pytest.approx(1e-5 + 9e-5 * 0.999 ** 1000)
    # one cycle: first cycle triangular, second cycle anneals below the base rate
    assert f(1000, 1e-5, 1e-4, 1000, mode="triangular", one_cycle=True) == pytest.approx(1e-4)
    assert f(3000, 1e-5, 1e-4, 1000, mode="triangular", one_cycle=True, annealing_factor=1e-3) == pytest.approx(1e-5 * 1e-3)
    assert f(2500, 1e-5, 1e-4, 1000, mode="triangular", one_cycle=True, annealing_factor=1e-3) == pytest.approx(
        1e-5 - 0.5 * (1e-5 - 1e-8))
    with pytest.raises(ValueError):
        f(0, 1e-5, 1e-4, 1000, mode="sawtooth")
    assert TS.learning_rate(TS.CLR_SCHEDULE, 2000, {"clr_stepsize": 2000, "clr_mode": "triangular"}) == pytest.approx(1e-4)


def test_exponential_policies():
    assert TS.exponentially_increasing_lr(0, 1e-10, 1.0, 10000) == pytest.approx(1e-10)
    assert TS.exponentially_increasing_lr(10000, 1e-10, 1.0, 10000) == pytest.approx(1.0)
    assert TS.exponentially_increasing_lr(5000, 1e-10, 1.0, 10000) == pytest.approx(1e-5)
    assert TS.exponentially_decreasing_lr(5000, 1e-6, 1e-2, 10000) == pytest.approx(1e-4)
    assert TS.learning_rate(TS.LR_RANGE_TEST, 4919) == pytest.approx(1e-10 * (1e10) ** (4919 / 9838))
