"""Flow-field IO, visualisation and the EPE metric of the CLI outputs
(the subset of /root/reference src/flowlib.py that ``Net.test`` uses:
read_flow :85-100, write_flow :126-142, flow_to_image :493-539, compute_color
:766-807, make_color_wheel :810-861, EPE of flow_error_mask :485-487).
Host-side NumPy: file formats, not arithmetic on the hot path."""
import numpy as np

TAG_FLOAT = 202021.25
UNKNOWN_FLOW_THRESH = 1e9


def read_flow(filename):
    """Middlebury .flo -> (H, W, 2) float32."""
    with open(filename, "rb") as f:
        head = np.frombuffer(f.read(12), dtype=np.uint8)
        if head.size != 12 or head[:4].view(np.float32)[0] != np.float32(TAG_FLOAT):
            raise ValueError("Magic number incorrect. Invalid .flo file")
        w, h = (int(v) for v in head[4:].view(np.int32))
        data = np.fromfile(f, np.float32, count=2 * w * h)
    if data.size != 2 * w * h:
        raise ValueError("truncated .flo file")
    return data.reshape(h, w, 2)


def write_flow(flow, filename):
    """(H, W, 2) -> .flo: f32 tag, i32 width, i32 height, row-major (u, v) float32."""
    flow = np.ascontiguousarray(flow, dtype=np.float32)
    if flow.ndim != 3 or flow.shape[2] != 2:
        raise ValueError("flow must be (H, W, 2)")
    h, w = flow.shape[:2]
    with open(filename, "wb") as f:
        np.array([TAG_FLOAT], np.float32).tofile(f)
        np.array([w, h], np.int32).tofile(f)
        flow.tofile(f)


def endpoint_error(flow, gt):
    """Mean end-point error over pixels whose ground truth is known (|gt| <= 1e9)."""
    flow = np.asarray(flow, np.float64)
    gt = np.asarray(gt, np.float64)
    known = (np.abs(gt[..., 0]) <= UNKNOWN_FLOW_THRESH) & (np.abs(gt[..., 1]) <= UNKNOWN_FLOW_THRESH)
    err = np.hypot(gt[..., 0] - flow[..., 0], gt[..., 1] - flow[..., 1])
    return float(err[known].mean())


_SEGMENTS = (("RY", 15), ("YG", 6), ("GC", 4), ("CB", 11), ("BM", 13), ("MR", 6))


def make_color_wheel():
    """55 x 3 Middlebury colour wheel."""
    n = sum(s for _, s in _SEGMENTS)
    wheel = np.zeros((n, 3))
    # (channel ramping up or down, channel held at 255) per segment
    plan = ((1, +1, 0), (0, -1, 1), (2, +1, 1), (1, -1, 2), (0, +1, 2), (2, -1, 0))
    row = 0
    for (_, size), (ramp_ch, direction, full_ch) in zip(_SEGMENTS, plan):
        ramp = np.floor(255 * np.arange(size) / size)
        wheel[row:row + size, full_ch] = 255
        wheel[row:row + size, ramp_ch] = ramp if direction > 0 else 255 - ramp
        row += size
    return wheel


def compute_color(u, v):
    """Colour-code a normalised flow (|flow| <= 1 inside the wheel), keeping the
    reference's 1-based interpolation indices so images match it byte for byte."""
    u = np.array(u)
    v = np.array(v)
    bad = np.isnan(u) | np.isnan(v)
    u[bad] = 0
    v[bad] = 0
    wheel = make_color_wheel()
    ncols = wheel.shape[0]
    rad = np.sqrt(u ** 2 + v ** 2)
    fk = (np.arctan2(-v, -u) / np.pi + 1) / 2 * (ncols - 1) + 1
    k0 = np.floor(fk).astype(int)
    k1 = k0 + 1
    k1[k1 == ncols + 1] = 1
    f = fk - k0
    img = np.zeros(u.shape + (3,))
    inside = rad <= 1
    for ch in range(3):
        col = (1 - f) * (wheel[k0 - 1, ch] / 255) + f * (wheel[k1 - 1, ch] / 255)
        col[inside] = 1 - rad[inside] * (1 - col[inside])
        col[~inside] *= 0.75
        img[:, :, ch] = np.uint8(np.floor(255 * col * (1 - bad)))
    return img


def flow_to_image(flow, maxflow=-1):
    """(H, W, 2) flow -> uint8 (H, W, 3); ``maxflow`` > 0 fixes the normalising radius."""
    u = np.array(flow[:, :, 0])
    v = np.array(flow[:, :, 1])
    unknown = (np.abs(u) > UNKNOWN_FLOW_THRESH) | (np.abs(v) > UNKNOWN_FLOW_THRESH)
    u[unknown] = 0
    v[unknown] = 0
    maxrad = np.max(np.sqrt(u ** 2 + v ** 2))
    if maxflow > 0:
        maxrad = maxflow
    if maxrad == 0:
        maxrad = 1
    eps = np.finfo(float).eps
    img = compute_color(u / (maxrad + eps), v / (maxrad + eps))
    img[np.repeat(unknown[:, :, None], 3, axis=2)] = 0
    return np.uint8(img)


# ---------------------------------------------------------------------------------------------------
# MPI-Sintel error metrics used by Net.test_batch (reference src/flowlib.py: flow_error :379-430,
# flow_error_mask :433-490, compute_all_metrics :215-375, get_metrics :182-212).  The reference indexes with
# one-element lists of boolean arrays, which the NumPy it was written for read as plain boolean masks;
# that reading is what is implemented here (tests/golden/metrics_golden.npz holds the reference's outputs).
# Arithmetic stays in the dtype of the inputs (float32 for .flo data), like the reference's.
# ---------------------------------------------------------------------------------------------------
def _angular_and_epe(stu, stv, su, sv, sel, clamp_ge):
    isu, isv = su[sel], sv[sel]
    an = 1.0 / np.sqrt(isu ** 2 + isv ** 2 + 1)
    istu, istv = stu[sel], stv[sel]
    tn = 1.0 / np.sqrt(istu ** 2 + istv ** 2 + 1)
    angle = (isu * an) * (istu * tn) + (isv * an) * (istv * tn) + (an * tn)
    angle[(angle >= 1.0) if clamp_ge else (angle == 1.0)] = 0.999
    with np.errstate(invalid="ignore"):
        ang = np.arccos(angle)
    mang = np.mean(ang) * 180 / np.pi
    stdang = np.std(ang * 180 / np.pi)
    epe = np.sqrt((stu - su) ** 2 + (stv - sv) ** 2)[sel]
    return mang, stdang, np.mean(epe)


def flow_error(tu, tv, u, v):
    """(mean angular error [deg], its std, mean EPE) over pixels with non-zero known ground truth.  Like the
    reference it zeroes the unknown pixels of its arguments IN PLACE and only clamps angle == 1.0 exactly, so
    rounding above 1 gives NaN angular statistics (flowlib.py:398-430)."""
    unknown = (abs(tu) > UNKNOWN_FLOW_THRESH) | (abs(tv) > UNKNOWN_FLOW_THRESH)
    for a in (tu, tv, u, v):
        a[unknown] = 0
    sel = (np.absolute(tu) > 0.0) | (np.absolute(tv) > 0.0)
    return _angular_and_epe(tu, tv, u, v, sel, clamp_ge=False)


def flow_error_mask(tu, tv, u, v, mask=None, gt_value=False, bord=0):
    """Same statistics over the pixels whose ground truth is known and whose `mask` entry differs from
    `gt_value` (mask=None: every known pixel) (flowlib.py:433-490)."""
    unknown = (abs(tu) > UNKNOWN_FLOW_THRESH) | (abs(tv) > UNKNOWN_FLOW_THRESH) | (mask == gt_value)
    sel = (unknown < 1) & ((abs(tu) >= 0.0) | (abs(tv) >= 0.0))
    return _angular_and_epe(tu, tv, u, v, sel, clamp_ge=True)


def compute_all_metrics(est_flow, gt_flow, occ_mask=None, inv_mask=None):
    """EPEall / EPEmat / EPEumat, S0-10 / S10-40 / S40+ and the angular statistics of MPI-Sintel for one
    frame (flowlib.py:215-375).  Masks are uint8 images, 255 = occluded / invalid.  Returns (metrics dict,
    not_occluded, s0_10_is_zero, s10_40_is_zero, s40plus_is_zero) -- the flags count frames in which a class is
    empty, for averaging over a sequence."""
    height, width, _ = gt_flow.shape
    gx, gy = gt_flow[:, :, 0], gt_flow[:, :, 1]
    ex, ey = est_flow[:, :, 0], est_flow[:, :, 1]
    occ = (occ_mask == 255) if occ_mask is not None else np.full((height, width), False)
    inv = (inv_mask == 255) if inv_mask is not None else np.full((height, width), False)
    m = {}
    m["mangall"], m["stdangall"], m["EPEall"] = flow_error_mask(gx, gy, ex, ey, inv, True)
    if occ.size and np.unique(occ).shape[0] > 1:
        m["mangmat"], m["stdangmat"], m["EPEmat"] = flow_error_mask(gx, gy, ex, ey, occ | inv, True)
        m["mangumat"], m["stdangumat"], m["EPEumat"] = flow_error_mask(gx, gy, ex, ey, occ & ~inv, False)
        not_occluded = 0
    else:
        m["mangmat"], m["stdangmat"], m["EPEmat"] = m["mangall"], m["stdangall"], m["EPEall"]
        m["mangumat"] = m["stdangumat"] = m["EPEumat"] = 0
        not_occluded = 1
    speed = np.sqrt(gx ** 2 + gy ** 2)
    cls = np.where(speed < 10, 0, np.where(speed <= 40, 1, 2))
    empties = []
    for c, key in ((0, "S0-10"), (1, "S10-40"), (2, "S40plus")):
        if np.any(cls == c):
            m[key] = flow_error_mask(gx, gy, ex, ey, (cls == c) & ~inv, False)[2]
            empties.append(0)
        else:
            m[key] = 0
            empties.append(1)
    return (m, not_occluded) + tuple(empties)


def get_metrics(metrics, average=False, flow_fname=None):
    """The 13-line text block the reference logs per frame / per sequence (flowlib.py:182-212)."""
    dash, line = "-" * 50, "_" * 50
    if average:
        title, name = "MPI-Sintel Flow Error Metrics (AVERAGE)", "For all files above"
    else:
        title, name = "MPI-Sintel Flow Error Metrics", flow_fname if flow_fname is not None else "Unknown filename"
    row = "{:<5s}{:^15.4f}{:^15.4f}{:^15.4f}".format
    head = "{:<5s}{:^15s}{:^15s}{:^15s}".format
    rows = [line, "{:^50}".format(title), "{:^50}".format(name), line, head("Mask", "MANG", "STDANG", "MEPE"), dash,
            row("(all)", metrics["mangall"], metrics["stdangall"], metrics["EPEall"]),
            row("(mat)", metrics["mangmat"], metrics["stdangmat"], metrics["EPEmat"]),
            row("(umt)", metrics["mangumat"], metrics["stdangumat"], metrics["EPEumat"]),
            line, head("", "S0-10", "S10-40", "S40+"), dash,
            row("(dis)", metrics["S0-10"], metrics["S10-40"], metrics["S40plus"])]
    return "\n".join(rows) + "\n"


def evaluate_flow(gt_flow, pred_flow):
    """Mean EPE of two (H, W, 2) fields (flowlib.py:634-647)."""
    return flow_error(gt_flow[:, :, 0].copy(), gt_flow[:, :, 1].copy(), pred_flow[:, :, 0].copy(),
                      pred_flow[:, :, 1].copy())[2]


def evaluate_flow_file(gt, pred):
    """Mean EPE of two .flo files (flowlib.py:619-631)."""
    return evaluate_flow(read_flow(gt), read_flow(pred))
