"""Oracle (test infrastructure): the three custom ops of the reference, on CPU.

PARITY UNPINNED: the reference kernels (src/ops/*/*.cu.cc) are GPU-only TF
plugins that cannot be built or run in this container, and the reference has no
known-answer vectors for them.  Each op is restated twice -- ``*_loops`` is a
literal per-thread transliteration of the kernel's indexing (pure-Python, tiny
inputs only), the unsuffixed function is vectorised NumPy -- and the two are
tested against each other (tests/test_oracle_ops.py).

All tensors NHWC float32, like the reference.
"""
import math

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------
# correlation
# ----------------------------------------------------------------------------
def correlation_geometry(H, W, kernel_size, max_displacement, stride_1, stride_2, pad):
    """Output geometry, following correlation_kernel.cc:38-61 (and the shape
    function correlation_op.cc:9-51).  Returns a dict."""
    if kernel_size % 2 == 0:
        raise ValueError("kernel_size must be odd")  # correlation_kernel.cc:23
    kr = (kernel_size - 1) // 2
    border = max_displacement + kr
    Hp, Wp = H + 2 * pad, W + 2 * pad
    # ceil of a float32 division, as written in the reference (:50-51)
    oh = int(math.ceil(float(F32(Hp - 2 * border) / F32(stride_1))))
    ow = int(math.ceil(float(F32(Wp - 2 * border) / F32(stride_1))))
    if oh < 1:
        raise ValueError("Neighborhood and kernel don't fit in input height.")  # :53
    if ow < 1:
        raise ValueError("Neighborhood and kernel don't fit in input width.")  # :55
    gr = max_displacement // stride_2
    gw = 2 * gr + 1
    return dict(kr=kr, border=border, Hp=Hp, Wp=Wp, oh=oh, ow=ow, gr=gr, gw=gw, D=gw * gw)


def _check_rank4_pair(a, b):
    if a.ndim != 4:
        raise ValueError("input_a must have rank 4")  # correlation_kernel.cc:31
    if b.ndim != 4:
        raise ValueError("input_b must have rank 4")  # :32
    if a.shape != b.shape:
        raise ValueError("input_a and input_b must have the same shape")  # op.cc:17 Merge


def _check_window(g, k, md, s1, s2):
    """The reference never bounds-checks the displaced read (it is in range for
    the call site pad == max_displacement); anything else reads outside the
    padded buffer = undefined behaviour.  The restatement refuses it."""
    lo = md - g["gr"] * s2
    hi_y = (g["oh"] - 1) * s1 + md + g["gr"] * s2 + k - 1
    hi_x = (g["ow"] - 1) * s1 + md + g["gr"] * s2 + k - 1
    if lo < 0 or hi_y >= g["Hp"] or hi_x >= g["Wp"]:
        raise ValueError("displacement window leaves the padded input (undefined in the reference)")


def correlation(a, b, kernel_size, max_displacement, stride_1, stride_2, pad):
    """Vectorised restatement of CorrelateData (correlation_kernel.cu.cc:45-110)
    on inputs zero-padded as Pad does (pad.cu.cc:46-74).  float64 accumulation,
    float32 result."""
    a = np.asarray(a, F32)
    b = np.asarray(b, F32)
    _check_rank4_pair(a, b)
    N, H, W, C = a.shape
    k, md, s1, s2 = kernel_size, max_displacement, stride_1, stride_2
    g = correlation_geometry(H, W, k, md, s1, s2, pad)
    _check_window(g, k, md, s1, s2)
    oh, ow, gr, gw, D = g["oh"], g["ow"], g["gr"], g["gw"], g["D"]
    A0 = np.pad(a, ((0, 0), (pad, pad), (pad, pad), (0, 0))).astype(np.float64)
    B0 = np.pad(b, ((0, 0), (pad, pad), (pad, pad), (0, 0))).astype(np.float64)
    ys = md + np.arange(oh) * s1  # y1, :46
    xs = md + np.arange(ow) * s1  # x1, :45
    out = np.zeros((N, oh, ow, D), np.float64)
    for d in range(D):
        s2o = (d % gw - gr) * s2  # :75
        s2p = (d // gw - gr) * s2  # :76
        acc = np.zeros((N, oh, ow), np.float64)
        for j in range(k):
            for i in range(k):
                pa = A0[:, ys + j][:, :, xs + i]
                pb = B0[:, ys + j + s2p][:, :, xs + i + s2o]
                acc += np.einsum("nyxc,nyxc->nyx", pa, pb)
        out[..., d] = acc
    return (out / float(k * k * C)).astype(F32)  # /sumelems, :105-110


def correlation_loops(a, b, kernel_size, max_displacement, stride_1, stride_2, pad):
    """Literal transliteration: one (x, y, n) block of 32 'threads' striding the
    channels, per-thread partial sums, thread 0 adds the 32 partials
    (correlation_kernel.cu.cc:21-119).  float32 arithmetic.  Tiny inputs only."""
    a = np.asarray(a, F32)
    b = np.asarray(b, F32)
    _check_rank4_pair(a, b)
    N, H, W, C = a.shape
    k, md, s1, s2 = kernel_size, max_displacement, stride_1, stride_2
    g = correlation_geometry(H, W, k, md, s1, s2, pad)
    _check_window(g, k, md, s1, s2)
    A0 = np.pad(a, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    B0 = np.pad(b, ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    out = np.zeros((N, g["oh"], g["ow"], g["D"]), F32)
    T = 32
    for n in range(N):
        for by in range(g["oh"]):
            for bx in range(g["ow"]):
                x1 = bx * s1 + md
                y1 = by * s1 + md
                for oc in range(g["D"]):
                    s2o = (oc % g["gw"] - g["gr"]) * s2
                    s2p = (oc // g["gw"] - g["gr"]) * s2
                    part = np.zeros(T, F32)
                    for j in range(k):
                        for i in range(k):
                            va = A0[n, y1 + j, x1 + i]
                            vb = B0[n, y1 + s2p + j, x1 + s2o + i]
                            for t in range(T):
                                for ch in range(t, C, T):
                                    part[t] = F32(part[t] + F32(va[ch] * vb[ch]))
                    total = F32(0)
                    for t in range(T):
                        total = F32(total + part[t])
                    out[n, by, bx, oc] = F32(total / F32(k * k * C))
    return out


def _ceil_div_roundoff(num, s1):
    # (num + ROUND_OFF*s1 - 1) / s1 + 1 - ROUND_OFF with C integer division
    # (correlation_grad_kernel.cu.cc:5,51-63): equals ceil(num / s1) for any sign.
    return -((-num) // s1)


def correlation_grad_loops(grad, a, b, kernel_size, max_displacement, stride_1, stride_2, pad):
    """CorrelateDataBackward0 / Backward1 (correlation_grad_kernel.cu.cc:20-101,
    :103-189) thread by thread.  Returns (dA, dB).  Small inputs only."""
    a = np.asarray(a, F32)
    b = np.asarray(b, F32)
    grad = np.asarray(grad, F32)
    N, H, W, C = a.shape
    k, md, s1, s2 = kernel_size, max_displacement, stride_1, stride_2
    g = correlation_geometry(H, W, k, md, s1, s2, pad)
    kr, gr, gw, oh, ow = g["kr"], g["gr"], g["gw"], g["oh"], g["ow"]
    A0 = np.pad(a, ((0, 0), (pad, pad), (pad, pad), (0, 0))).astype(np.float64)
    B0 = np.pad(b, ((0, 0), (pad, pad), (pad, pad), (0, 0))).astype(np.float64)
    G = grad.astype(np.float64)
    sumelems = float((2 * kr + 1) ** 2 * C)
    dA = np.zeros((N, H, W, C), np.float64)
    dB = np.zeros((N, H, W, C), np.float64)
    for n in range(N):
        for yy in range(H):
            for xx in range(W):
                x = xx + pad
                y = yy + pad
                # ---- grad wrt A (:51-96)
                xmin = _ceil_div_roundoff(x - 2 * kr - md, s1)
                ymin = _ceil_div_roundoff(y - 2 * kr - md, s1)
                xmax = (x - md) // s1
                ymax = (y - md) // s1
                if xmax >= 0 and ymax >= 0 and xmin <= ow - 1 and ymin <= oh - 1:
                    xmin_, xmax_ = max(0, xmin), min(ow - 1, xmax)
                    ymin_, ymax_ = max(0, ymin), min(oh - 1, ymax)
                    for p in range(-gr, gr + 1):
                        for o in range(-gr, gr + 1):
                            yb, xb = y + s2 * p, x + s2 * o
                            if not (0 <= yb < g["Hp"] and 0 <= xb < g["Wp"]):
                                raise ValueError("displacement window leaves the padded input")
                            op = (p + gr) * gw + (o + gr)
                            gs = G[n, ymin_:ymax_ + 1, xmin_:xmax_ + 1, op].sum()
                            dA[n, yy, xx, :] += gs * B0[n, yb, xb, :]
                # ---- grad wrt B (:126-187)
                for p in range(-gr, gr + 1):
                    for o in range(-gr, gr + 1):
                        s2o, s2p = s2 * o, s2 * p
                        xmin = _ceil_div_roundoff(x - 2 * kr - md - s2o, s1)
                        ymin = _ceil_div_roundoff(y - 2 * kr - md - s2p, s1)
                        xmax = (x - md - s2o) // s1
                        ymax = (y - md - s2p) // s1
                        if xmax >= 0 and ymax >= 0 and xmin <= ow - 1 and ymin <= oh - 1:
                            xmin_, xmax_ = max(0, xmin), min(ow - 1, xmax)
                            ymin_, ymax_ = max(0, ymin), min(oh - 1, ymax)
                            ya, xa = y - s2p, x - s2o
                            if not (0 <= ya < g["Hp"] and 0 <= xa < g["Wp"]):
                                raise ValueError("displacement window leaves the padded input")
                            op = (p + gr) * gw + (o + gr)
                            gs = G[n, ymin_:ymax_ + 1, xmin_:xmax_ + 1, op].sum()
                            dB[n, yy, xx, :] += gs * A0[n, ya, xa, :]
    return (dA / sumelems).astype(F32), (dB / sumelems).astype(F32)


def correlation_grad(grad, a, b, kernel_size, max_displacement, stride_1, stride_2, pad):
    """Vectorised form of the same two kernels for kernel_size == 1 and
    stride_1 == 1 (the only configuration the model uses, flownet_c.py:40), where
    the clamped (y, x) range collapses to the single output pixel under the
    input pixel.  Falls back to the loop form otherwise."""
    k, md, s1, s2 = kernel_size, max_displacement, stride_1, stride_2
    if k != 1 or s1 != 1:
        return correlation_grad_loops(grad, a, b, k, md, s1, s2, pad)
    a = np.asarray(a, F32)
    b = np.asarray(b, F32)
    N, H, W, C = a.shape
    g = correlation_geometry(H, W, k, md, s1, s2, pad)
    gr, gw, oh, ow = g["gr"], g["gw"], g["oh"], g["ow"]
    G = np.asarray(grad, np.float64)
    A0 = np.pad(a, ((0, 0), (pad, pad), (pad, pad), (0, 0))).astype(np.float64)
    B0 = np.pad(b, ((0, 0), (pad, pad), (pad, pad), (0, 0))).astype(np.float64)
    dA0 = np.zeros_like(A0)
    dB0 = np.zeros_like(B0)
    ys = md + np.arange(oh)
    xs = md + np.arange(ow)
    for d in range(g["D"]):
        s2o = (d % gw - gr) * s2
        s2p = (d // gw - gr) * s2
        gd = G[..., d][..., None]
        dA0[:, md:md + oh, md:md + ow] += gd * B0[:, ys + s2p][:, :, xs + s2o]
        dB0[:, md + s2p:md + s2p + oh, md + s2o:md + s2o + ow] += gd * A0[:, ys][:, :, xs]
    sl = slice(pad, pad + H), slice(pad, pad + W)
    return ((dA0[:, sl[0], sl[1]] / C).astype(F32), (dB0[:, sl[0], sl[1]] / C).astype(F32))


# ----------------------------------------------------------------------------
# flow_warp
# ----------------------------------------------------------------------------
def _check_warp_args(image, flow):
    if image.ndim != 4:
        raise ValueError("Input images must have rank 4")  # flow_warp.cc:22
    if flow.ndim != 4:
        raise ValueError("Input flow must have rank 4")  # :23
    if image.shape[:3] != flow.shape[:3]:
        raise ValueError("Input images and flows must have same N, H, W")  # :24-29
    if flow.shape[3] != 2:
        raise ValueError("Input flow must have 2 channels")  # :30


def _warp_coords(flow):
    N, H, W, _ = flow.shape
    xg = np.arange(W, dtype=F32)[None, None, :]
    yg = np.arange(H, dtype=F32)[None, :, None]
    with np.errstate(invalid="ignore"):
        x2 = (xg + flow[..., 0]).astype(F32)  # flow_warp.cu.cc:45
        y2 = (yg + flow[..., 1]).astype(F32)  # :46
        valid = (x2 >= 0) & (y2 >= 0) & (x2 < F32(W)) & (y2 < F32(H))  # :80 (NaN fails)
    x2v = np.where(valid, x2, F32(0))
    y2v = np.where(valid, y2, F32(0))
    xL = x2v.astype(np.int64)  # int(x2): truncation, :54
    yT = y2v.astype(np.int64)
    xR = np.minimum(xL + 1, W - 1)  # :56
    yB = np.minimum(yT + 1, H - 1)
    alpha = (x2v - xL.astype(F32)).astype(F32)  # :64
    beta = (y2v - yT.astype(F32)).astype(F32)
    return valid, x2v, y2v, xL, xR, yT, yB, alpha, beta


def flow_warp(image, flow):
    """FlowWarpKernel (flow_warp.cu.cc:44-95): backward bilinear warp, zero
    outside the image, clamped right/bottom neighbour.  float32 arithmetic."""
    image = np.asarray(image, F32)
    flow = np.asarray(flow, F32)
    _check_warp_args(image, flow)
    N, H, W, C = image.shape
    valid, _, _, xL, xR, yT, yB, alpha, beta = _warp_coords(flow)
    one = F32(1)
    cTL = ((one - alpha) * (one - beta)).astype(F32)[..., None]  # :66-69
    cTR = (alpha * (one - beta)).astype(F32)[..., None]
    cBL = ((one - alpha) * beta).astype(F32)[..., None]
    cBR = (alpha * beta).astype(F32)[..., None]
    n = np.arange(N)[:, None, None]
    out = (cTL * image[n, yT, xL] + cTR * image[n, yT, xR]
           + cBL * image[n, yB, xL] + cBR * image[n, yB, xR]).astype(F32)  # :82-86
    return np.where(valid[..., None], out, F32(0)).astype(F32)


def flow_warp_loops(image, flow):
    """Per-pixel scalar form of the same kernel (tiny inputs)."""
    image = np.asarray(image, F32)
    flow = np.asarray(flow, F32)
    _check_warp_args(image, flow)
    N, H, W, C = image.shape
    out = np.zeros_like(image)
    for n in range(N):
        for y in range(H):
            for x in range(W):
                x2 = F32(F32(x) + flow[n, y, x, 0])
                y2 = F32(F32(y) + flow[n, y, x, 1])
                if not (x2 >= 0 and y2 >= 0 and x2 < W and y2 < H):
                    continue
                xL, yT = int(x2), int(y2)
                xR, yB = min(xL + 1, W - 1), min(yT + 1, H - 1)
                al, be = F32(x2 - F32(xL)), F32(y2 - F32(yT))
                one = F32(1)
                for c in range(C):
                    out[n, y, x, c] = (F32((one - al) * (one - be)) * image[n, yT, xL, c]
                                       + F32(al * (one - be)) * image[n, yT, xR, c]
                                       + F32((one - al) * be) * image[n, yB, xL, c]
                                       + F32(al * be) * image[n, yB, xR, c])
    return out


def flow_warp_grad(image, flow, grad):
    """FlowWarpGradKernel (flow_warp_grad.cu.cc:30-86).  Returns
    (image_grad, flow_grad); out-of-range pixels contribute nothing and get a
    zero flow gradient (memsets, :107-108).  float64 scatter accumulation (the
    reference's atomicAdd order is not deterministic)."""
    image = np.asarray(image, F32)
    flow = np.asarray(flow, F32)
    grad = np.asarray(grad, F32)
    _check_warp_args(image, flow)
    N, H, W, C = image.shape
    valid, x2, y2, xL, xR, yT, yB, alpha, beta = _warp_coords(flow)
    g = np.where(valid[..., None], grad, F32(0)).astype(np.float64)
    al, be = alpha.astype(np.float64)[..., None], beta.astype(np.float64)[..., None]
    dI = np.zeros(image.shape, np.float64)
    n = np.broadcast_to(np.arange(N)[:, None, None], valid.shape)
    np.add.at(dI, (n, yT, xL), g * (1 - al) * (1 - be))  # :43-52
    np.add.at(dI, (n, yT, xR), g * al * (1 - be))
    np.add.at(dI, (n, yB, xL), g * (1 - al) * be)
    np.add.at(dI, (n, yB, xR), g * al * be)
    I = image.astype(np.float64)
    TL, TR, BL, BR = I[n, yT, xL], I[n, yT, xR], I[n, yB, xL], I[n, yB, xR]
    gy = (yB.astype(F32) - y2).astype(np.float64)[..., None]  # gamma = iy2_B - y2, :54
    gx = (xR.astype(F32) - x2).astype(np.float64)[..., None]  # gamma = ix2_R - x2, :71
    du = (g * (gy * (TR - TL) + (1 - gy) * (BR - BL))).sum(-1)  # :57-69
    dv = (g * (gx * (BL - TL) + (1 - gx) * (BR - TR))).sum(-1)  # :74-85
    dflow = np.stack([du, dv], -1)
    dflow = np.where(valid[..., None], dflow, 0.0)
    return dI.astype(F32), dflow.astype(F32)


# ----------------------------------------------------------------------------
# downsample
# ----------------------------------------------------------------------------
def _round_half_away(x):
    # C round(): half away from zero (downsample_kernel_gpu.cu.cc:44-45)
    return int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))


def _check_downsample_args(x, size):
    if x.ndim != 4:
        raise ValueError("Input images must have rank 4")  # downsample_kernel.cc:25
    if len(size) != 2:
        raise ValueError("size must have 2 elements")  # :19


def downsample_loops(x, size):
    """DownsampleKernel thread by thread (downsample_kernel_gpu.cu.cc:35-76),
    float32, same loop order.  Small inputs only."""
    x = np.asarray(x, F32)
    _check_downsample_args(x, size)
    N, Hin, Win, C = x.shape
    oh, ow = int(size[0]), int(size[1])
    with np.errstate(divide="ignore", invalid="ignore"):
        wscale = F32(Win - 1) / F32(ow - 1)  # :92
        hscale = F32(Hin - 1) / F32(oh - 1)  # :93
    wr, hr = int(math.ceil(wscale)), int(math.ceil(hscale))  # :95-96
    out = np.zeros((N, oh, ow, C), F32)
    for n in range(N):
        for dy in range(oh):
            for dx in range(ow):
                srcx = F32(F32(F32(dx) / F32(ow - 1)) * F32(Win - 1))  # :41
                srcy = F32(F32(F32(dy) / F32(oh - 1)) * F32(Hin - 1))  # :42
                ix, iy = _round_half_away(float(srcx)), _round_half_away(float(srcy))
                for c in range(C):
                    av, aw, an = F32(0), F32(0), F32(0)
                    for yy in range(iy - hr, iy + hr + 1):
                        for xx in range(ix - wr, ix + wr + 1):
                            if 0 <= xx < Win and 0 <= yy < Hin:
                                s = x[n, yy, xx, c]
                                w = F32(max(F32(0), F32(F32(1) - F32(abs(F32(xx) - srcx)) / wscale))
                                        * max(F32(0), F32(F32(1) - F32(abs(F32(yy) - srcy)) / hscale)))
                                if s != s:
                                    an = F32(an + w)
                                    s, w = F32(0), F32(0)
                                av = F32(av + F32(s * w))
                                aw = F32(aw + w)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        out[n, dy, dx, c] = F32(np.nan) if F32(an) / F32(aw) > 0.5 else F32(av / aw)
    return out


def _tri_weights(n_in, n_out):
    """[n_out, n_in] float64 matrix of max(0, 1-|i-src|/scale) restricted to the
    kernel's window [round(src)-ceil(scale), round(src)+ceil(scale)] and to the
    image; src and scale computed in float32 exactly as the kernel does."""
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = F32(n_in - 1) / F32(n_out - 1)
    r = int(math.ceil(scale))
    Wm = np.zeros((n_out, n_in), np.float64)
    for d in range(n_out):
        src = F32(F32(F32(d) / F32(n_out - 1)) * F32(n_in - 1))
        ic = _round_half_away(float(src))
        lo, hi = max(0, ic - r), min(n_in - 1, ic + r)
        idx = np.arange(lo, hi + 1)
        w = F32(1) - np.abs(idx.astype(F32) - src).astype(F32) / scale
        Wm[d, lo:hi + 1] = np.maximum(F32(0), w.astype(F32))
    return Wm


def downsample(x, size):
    """Separable, vectorised form of DownsampleKernel: the tap weight is a
    product wx(xo)*wy(yo) (downsample_kernel_gpu.cu.cc:57-58), so the three
    accumulators (value, weight, NaN weight; :59-65) are each Wy @ F @ Wx^T.
    float64 accumulation, float32 result.  NaN-aware exactly as :68-72."""
    x = np.asarray(x, F32)
    _check_downsample_args(x, size)
    N, Hin, Win, C = x.shape
    oh, ow = int(size[0]), int(size[1])
    Wy = _tri_weights(Hin, oh)
    Wx = _tri_weights(Win, ow)
    isnan = np.isnan(x)
    val = np.where(isnan, 0.0, x).astype(np.float64)
    ok = (~isnan).astype(np.float64)
    nn = isnan.astype(np.float64)

    def sep(t):
        return np.einsum("ah,nhwc,bw->nabc", Wy, t, Wx, optimize=True)

    av, aw, an = sep(val), sep(ok), sep(nn)
    with np.errstate(divide="ignore", invalid="ignore"):
        res = av / aw
        res = np.where(an / aw > 0.5, np.nan, res)
    return res.astype(F32)
