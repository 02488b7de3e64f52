"""Training schedules and learning-rate policies of the reference (/root/reference src/training_schedules.py:14-188,
src/utils.py:24-135, src/net.py:1139-1207), as data and plain host functions of the global step -- the trainer asks
`learning_rate(schedule, step, params)` once per step and passes the value to the Adam kernel.

Piecewise-constant schedules: `step_values` are the boundaries of tf.train.piecewise_constant (net.py:1205-1207:
rate i applies while boundary i-1 < step <= boundary i).
The PWC-Net+-style fine-tuning stages (training_schedules.py:78-181) follow one pattern -- ten boundaries at fixed
offsets inside a 150k-step stage, the rate halved at each -- and are generated from (stage, first rate)."""

_ADAM = {'momentum': 0.9, 'momentum2': 0.999}


def _piecewise(step_values, learning_rates, max_iters, l2=0.0004):
    return dict(step_values=list(step_values), learning_rates=list(learning_rates), l2_regularization=l2,
                max_iters=max_iters, **_ADAM)


# lmb-freiburg/flownet2 solver schedules (training_schedules.py:46-73, :182-188)
LONG_SCHEDULE = _piecewise([400000, 600000, 800000, 1000000], [1e-4, 5e-5, 2.5e-5, 1.25e-5, 6.25e-6], 1200000)
FINE_SCHEDULE = _piecewise([200000, 300000, 400000], [1e-5, 5e-6, 2.5e-6, 1.25e-6], 500000)
SHORT_SCHEDULE = _piecewise([300000, 400000, 500000], [1e-4, 5e-5, 2.5e-5, 1.25e-5], 600000)
FINETUNE_ROB = _piecewise([300000, 400000, 500000], [1e-4, 5e-5, 2.5e-5, 1.25e-5], 600000)

_STAGE_OFFSETS = (45000, 65000, 85000, 95000, 97500, 100000, 110000, 120000, 130000, 140000)


def _finetune_stage(stage, first_lr):
    base = 150000 * (stage - 1)
    return _piecewise([base + o for o in _STAGE_OFFSETS], [first_lr / 2 ** i for i in range(11)], base + 150000)


# learning-rate disruptions of PWC-Net+ fine-tuning (training_schedules.py:78-181): first rate of each stage
FINETUNE_SINTEL = {s: _finetune_stage(s, lr) for s, lr in ((1, 5e-5), (2, 3e-5), (3, 2e-5), (4, 1e-5), (5, 5e-6))}
FINETUNE_KITTI = {s: _finetune_stage(s, lr) for s, lr in ((1, 4e-5), (2, 4e-5), (3, 2e-5), (4, 1e-5))}

# policies computed from the step (training_schedules.py:14-44): 'learning_rates' names the policy
EXP_DECREASING = dict(learning_rates='exp_decr', l2_regularization=0.0, max_iters=10000, **_ADAM)
ONECYCLE_SCHEDULE = dict(learning_rates='one_cycle', l2_regularization=0.0, max_iters=30000, **_ADAM)
CLR_SCHEDULE = dict(learning_rates='clr', l2_regularization=0.0, max_iters=10000, **_ADAM)
LR_RANGE_TEST = dict(learning_rates='range_test', l2_regularization=0.0, max_iters=9838, **_ADAM)

SCHEDULES = {
    'long_schedule': LONG_SCHEDULE, 'fine_schedule': FINE_SCHEDULE, 'short_schedule': SHORT_SCHEDULE,
    'finetune_rob': FINETUNE_ROB, 'exp_decr': EXP_DECREASING, 'one_cycle': ONECYCLE_SCHEDULE, 'clr': CLR_SCHEDULE,
    'lr_range_test': LR_RANGE_TEST,
}
SCHEDULES.update({'finetune_sintel_s%d' % s: v for s, v in FINETUNE_SINTEL.items()})
SCHEDULES.update({'finetune_kitti_s%d' % s: v for s, v in FINETUNE_KITTI.items()})

# defaults of the policy parameters (the reference passes them as train.py arguments, net.py:1005-1190)
DEFAULT_PARAMS = {'clr_min_lr': 1e-5, 'clr_max_lr': 1e-4, 'clr_stepsize': 2000, 'clr_gamma': 0.99994,
                  'clr_mode': 'triangular2', 'one_cycle_annealing_factor': 1e-3, 'start_lr': 1e-10, 'end_lr': 1.0}


def exponentially_increasing_lr(step, min_lr=1e-10, max_lr=1.0, num_iters=10000):
    """utils.py:24-41 (the LR range test): min_lr * (max_lr / min_lr) ** (step / num_iters)."""
    return min_lr * (max_lr / min_lr) ** (step / num_iters)


def exponentially_decreasing_lr(step, min_lr=1e-10, max_lr=1.0, num_iters=10000):
    """utils.py:44-61: max_lr * (min_lr / max_lr) ** (step / num_iters)."""
    return max_lr * (min_lr / max_lr) ** (step / num_iters)


def cyclic_lr(step, base_lr, max_lr, step_size, gamma=0.99994, mode='triangular2', one_cycle=False,
              annealing_factor=1e-3):
    """utils.py:64-135 (L. N. Smith's cyclical rates + the one-cycle policy):
    cycle = floor(1 + step / (2 step_size)); x = |step / step_size - 2 cycle + 1|; a1 = max(0, 1 - x);
    lr = base + (max - base) * a1, the amplitude halved per cycle ('triangular2') or scaled by gamma**step
    ('exponential'); with one_cycle the SECOND cycle anneals instead: lr = base - base (1 - annealing_factor) a1."""
    if mode not in ('triangular', 'triangular2', 'exponential'):
        raise ValueError("mode must be 'triangular', 'triangular2' or 'exponential'")
    import math
    cycle = math.floor(1.0 + step / (2.0 * step_size))
    x = abs(step / step_size - 2.0 * cycle + 1.0)
    a1 = max(0.0, 1.0 - x)
    annealing = one_cycle and cycle == 2
    clr = a1 * ((base_lr - base_lr * annealing_factor) if annealing else (max_lr - base_lr))
    if mode == 'triangular2' and not one_cycle:
        clr /= float(2 ** int(cycle - 1))
    if mode == 'exponential' and not one_cycle:
        clr *= gamma ** step
    return base_lr - clr if annealing else base_lr + clr


def learning_rate(schedule, step, params=None):
    """The rate applied at global step `step` (0-based, as the reference's global_step tensor before the update)."""
    lrs = schedule['learning_rates']
    p = dict(DEFAULT_PARAMS, **(params or {}))
    if isinstance(lrs, str):
        kind = lrs.lower()
        if kind == 'clr':
            return cyclic_lr(step, p['clr_min_lr'], p['clr_max_lr'], p['clr_stepsize'], p['clr_gamma'], p['clr_mode'])
        if kind == 'one_cycle':
            return cyclic_lr(step, p['clr_min_lr'], p['clr_max_lr'], p['clr_stepsize'], mode='triangular', one_cycle=True,
                             annealing_factor=p['one_cycle_annealing_factor'])
        if kind == 'exp_decr':
            return exponentially_decreasing_lr(step, p['clr_min_lr'], p['clr_max_lr'], schedule['max_iters'])
        if kind == 'range_test':
            return exponentially_increasing_lr(step, p['start_lr'], p['end_lr'], schedule['max_iters'])
        raise ValueError("unknown learning-rate policy %r" % lrs)
    # tf.train.piecewise_constant(global_step, boundaries, values) (net.py:1205-1207): values[0] while step <=
    # boundaries[0], values[i] while boundaries[i-1] < step <= boundaries[i], values[-1] beyond the last boundary
    for i, b in enumerate(schedule['step_values']):
        if step <= b:
            return lrs[i]
    return lrs[-1]
