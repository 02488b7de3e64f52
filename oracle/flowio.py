"""Oracle (test infrastructure): .flo IO, EPE and Middlebury colour coding.

PINNED against the reference's own src/flowlib.py executed in the build
container (tests/golden/make_golden_flowlib.py) and against the reference's
sample files data/samples/{0,1}flow.flo (copied to tests/golden/samples/).
"""
import numpy as np

FLO_MAGIC = np.float32(202021.25)
UNKNOWN_FLOW_THRESH = 1e9  # flowlib.py:18


def read_flow(path):
    """flowlib.py:85-100 -- f32 magic 202021.25, i32 W, i32 H, H*W*2 f32."""
    with open(path, "rb") as f:
        magic = np.fromfile(f, np.float32, count=1)
        if magic.size != 1 or magic[0] != FLO_MAGIC:
            raise ValueError("Magic number incorrect. Invalid .flo file")
        w = int(np.fromfile(f, np.int32, count=1)[0])
        h = int(np.fromfile(f, np.int32, count=1)[0])
        data = np.fromfile(f, np.float32, count=2 * w * h)
    return data.reshape(h, w, 2)


def flo_bytes(flow):
    """flowlib.py:126-142 -- the exact byte image write_flow produces."""
    flow = np.ascontiguousarray(flow, np.float32)
    h, w = flow.shape[:2]
    return (FLO_MAGIC.tobytes() + np.int32(w).tobytes() + np.int32(h).tobytes() + flow.tobytes())


def write_flow(flow, path):
    with open(path, "wb") as f:
        f.write(flo_bytes(flow))


def mean_epe(flow, gt):
    """flowlib.py:485-487 restated without the numpy>=2-incompatible list
    indexing (SURVEY.md defect D8): mean over valid pixels of
    sqrt((tu-u)^2 + (tv-v)^2), valid = |gt| <= 1e9."""
    flow = np.asarray(flow, np.float64)
    gt = np.asarray(gt, np.float64)
    valid = (np.abs(gt[..., 0]) <= UNKNOWN_FLOW_THRESH) & (np.abs(gt[..., 1]) <= UNKNOWN_FLOW_THRESH)
    e = np.sqrt((gt[..., 0] - flow[..., 0]) ** 2 + (gt[..., 1] - flow[..., 1]) ** 2)
    return float(e[valid].mean())


def make_color_wheel():
    """flowlib.py:810-861 -- 55-entry Middlebury wheel."""
    RY, YG, GC, CB, BM, MR = 15, 6, 4, 11, 13, 6
    wheel = np.zeros((RY + YG + GC + CB + BM + MR, 3))
    col = 0
    wheel[0:RY, 0] = 255
    wheel[0:RY, 1] = np.floor(255 * np.arange(RY) / RY)
    col += RY
    wheel[col:col + YG, 0] = 255 - np.floor(255 * np.arange(YG) / YG)
    wheel[col:col + YG, 1] = 255
    col += YG
    wheel[col:col + GC, 1] = 255
    wheel[col:col + GC, 2] = np.floor(255 * np.arange(GC) / GC)
    col += GC
    wheel[col:col + CB, 1] = 255 - np.floor(255 * np.arange(CB) / CB)
    wheel[col:col + CB, 2] = 255
    col += CB
    wheel[col:col + BM, 2] = 255
    wheel[col:col + BM, 0] = np.floor(255 * np.arange(BM) / BM)
    col += BM
    wheel[col:col + MR, 2] = 255 - np.floor(255 * np.arange(MR) / MR)
    wheel[col:col + MR, 0] = 255
    return wheel


def compute_color(u, v):
    """flowlib.py:766-807 (including its 1-based k0/k1 indexing)."""
    u = np.array(u)  # dtype follows the caller, as in the reference
    v = np.array(v)
    nan_idx = np.isnan(u) | np.isnan(v)
    u[nan_idx] = 0
    v[nan_idx] = 0
    wheel = make_color_wheel()
    ncols = wheel.shape[0]
    rad = np.sqrt(u ** 2 + v ** 2)
    a = np.arctan2(-v, -u) / np.pi
    fk = (a + 1) / 2 * (ncols - 1) + 1
    k0 = np.floor(fk).astype(int)
    k1 = k0 + 1
    k1[k1 == ncols + 1] = 1
    f = fk - k0
    img = np.zeros(u.shape + (3,))
    for i in range(3):
        tmp = wheel[:, i]
        col0 = tmp[k0 - 1] / 255
        col1 = tmp[k1 - 1] / 255
        col = (1 - f) * col0 + f * col1
        idx = rad <= 1
        col[idx] = 1 - rad[idx] * (1 - col[idx])
        col[~idx] *= 0.75
        img[:, :, i] = np.uint8(np.floor(255 * col * (1 - nan_idx)))
    return img


def flow_to_image(flow, maxflow=-1):
    """flowlib.py:493-539."""
    u = np.array(flow[:, :, 0])  # no up-cast: the reference computes in the
    v = np.array(flow[:, :, 1])  # flow's own dtype (float32 from read_flow)
    unknown = (np.abs(u) > UNKNOWN_FLOW_THRESH) | (np.abs(v) > UNKNOWN_FLOW_THRESH)
    u[unknown] = 0
    v[unknown] = 0
    maxrad = np.max(np.sqrt(u ** 2 + v ** 2))  # np.max(np.max(rad), -1) in the reference
    if maxflow > 0:
        maxrad = maxflow
    if maxrad == 0:
        maxrad = 1
    eps = np.finfo(float).eps
    img = compute_color(u / (maxrad + eps), v / (maxrad + eps))
    img[np.repeat(unknown[:, :, None], 3, axis=2)] = 0
    return np.uint8(img)
