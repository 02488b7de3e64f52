"""Generate tests/golden/metrics_golden.npz by RUNNING the reference's own MPI-Sintel metric code
(src/flowlib.py: flow_error :379-430, flow_error_mask :433-490, compute_all_metrics :215-375,
get_metrics :182-212).  Build container only: needs /root/reference.

Two accommodations, neither touches the reference's files:
  * ``png`` / ``imageio`` are not installed and not used by these functions: placeholder modules;
  * the reference indexes with a one-element list holding a boolean array (``angle[[angle >= 1.0]]``), which
    NumPy < 1.23 read as the tuple ``(mask,)`` and NumPy 2 rejects.  The inputs are handed over as an ndarray
    subclass that restores that reading, so the reference's own statements execute unchanged.
Only inputs and the reference's outputs are stored.

    python tests/golden/make_golden_metrics.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class LegacyIndexArray(np.ndarray):
    """ndarray whose [list-of-one-array] index means [(array,)], as in the NumPy the reference was written for."""

    @staticmethod
    def _fix(idx):
        if isinstance(idx, list) and len(idx) == 1 and isinstance(idx[0], np.ndarray):
            return (np.asarray(idx[0]),)
        return idx

    def __getitem__(self, idx):
        return super().__getitem__(self._fix(idx))

    def __setitem__(self, idx, value):
        super().__setitem__(self._fix(idx), value)


def legacy(a):
    return np.array(a, copy=True).view(LegacyIndexArray)


def main():
    for m in ("png", "imageio"):
        mod = types.ModuleType(m)
        mod.imread = lambda *a, **k: None
        sys.modules[m] = mod
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    from src import flowlib  # the reference implementation

    out = {}
    rng = np.random.default_rng(11)
    gt = flowlib.read_flow(os.path.join(REF, "data/samples/0flow.flo"))[::4, ::4].copy()   # 96 x 128, real motion
    est = (gt + rng.standard_normal(gt.shape).astype(np.float32) * 0.7).astype(np.float32)
    est[5, 6] = gt[5, 6]                      # an exact match (angle == 1 branch)
    gt[20:23, 30:33] = 2e9                    # unknown ground truth
    occ = np.zeros(gt.shape[:2], np.uint8)
    occ[40:60, 50:90] = 255
    inv = np.zeros(gt.shape[:2], np.uint8)
    inv[0:4, :] = 255
    out.update(gt=gt, est=est, occ=occ, inv=inv)

    def run_all(tag, est_, gt_, occ_, inv_):
        m, not_occ, z0, z1, z2 = flowlib.compute_all_metrics(legacy(est_), legacy(gt_), occ_mask=occ_, inv_mask=inv_)
        keys = sorted(m)
        out[tag + "_keys"] = np.array(keys)
        out[tag + "_values"] = np.array([float(m[k]) for k in keys], np.float64)
        out[tag + "_counts"] = np.array([not_occ, z0, z1, z2])
        out[tag + "_text"] = np.array(flowlib.get_metrics(m, flow_fname="frame_0001"))
        return m

    m = run_all("full", est, gt, occ, inv)
    run_all("nomask", est, gt, None, None)
    small_gt = np.clip(gt, -3, 3)             # every displacement < 10: the S10-40 / S40+ "empty" branches
    run_all("small", est, small_gt, occ, None)
    out["avg_text"] = np.array(flowlib.get_metrics(m, average=True))
    # the two building blocks on their own
    mask = (occ == 255)
    r = flowlib.flow_error_mask(legacy(gt[..., 0]), legacy(gt[..., 1]), legacy(est[..., 0]), legacy(est[..., 1]),
                                mask, True, 0)
    out["fem_ignore_true"] = np.array(r, np.float64)
    r = flowlib.flow_error_mask(legacy(gt[..., 0]), legacy(gt[..., 1]), legacy(est[..., 0]), legacy(est[..., 1]),
                                mask, False, 0)
    out["fem_ignore_false"] = np.array(r, np.float64)
    r = flowlib.flow_error(legacy(gt[..., 0]), legacy(gt[..., 1]), legacy(est[..., 0]), legacy(est[..., 1]))
    out["flow_error"] = np.array(r, np.float64)
    np.savez_compressed(os.path.join(HERE, "metrics_golden.npz"), **out)
    print("wrote metrics_golden.npz:", {k: (v.shape if v.ndim else v.item()) for k, v in out.items() if "text" not in k and v.size < 20})


if __name__ == "__main__":
    main()
