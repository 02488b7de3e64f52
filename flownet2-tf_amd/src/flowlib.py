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
