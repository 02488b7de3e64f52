"""The schedule constants the hot path needs (/root/reference
src/training_schedules.py:46-53: LONG_SCHEDULE).  The other schedule dicts of the
reference are configuration data outside the hot-path scope."""

LONG_SCHEDULE = {
    'step_values': [400000, 600000, 800000, 1000000],
    'learning_rates': [0.0001, 0.00005, 0.000025, 0.0000125, 0.00000625],
    'momentum': 0.9,
    'momentum2': 0.999,
    'l2_regularization': 0.0004,
    'max_iters': 1200000,
}
