"""Oracle (test infrastructure only): CPU restatement of the reference's preprocessing plugin
(src/ops/preprocessing/kernels).  PARITY UNPINNED: the plugin needs TensorFlow headers to build and has no
known-answer vectors; two forms are kept (vectorised and literal per-pixel loops) and tested against each other.
  transmat_from_coeff / transmat_inverse   augmentation_base.cc:8-60 (TransMat::fromCoeff, ::inverse, ::leftMultiply)
  augment / augment_loops                  data_augmentation.cc:58-150 (spatial bilinear + chromatic chain)
  flow_augmentation / _loops               flow_augmentation.cc:30-66
  corners_fit                              augmentation_base.cc:313-345 (the validity test of the coefficients)
"""
import numpy as np

F32 = np.float32


def _left_multiply(t, u):
    t0, t1, t2, t3, t4, t5 = t
    u0, u1, u2, u3, u4, u5 = (F32(v) for v in u)
    return [t0 * u0 + t3 * u1, t1 * u0 + t4 * u1, t2 * u0 + t5 * u1 + u2,
            t0 * u3 + t3 * u4, t1 * u3 + t4 * u4, t2 * u3 + t5 * u4 + u5]


def transmat_from_coeff(coeff, out_w, out_h, src_w, src_h):
    """coeff: dict with any of dx, dy, angle, zoom_x, zoom_y (absent = 'has no value').  Returns 6 float32:
    output pixel -> source position.  Order of the factors as augmentation_base.cc:13-35."""
    t = [F32(1), F32(0), F32(0), F32(0), F32(1), F32(0)]
    t = _left_multiply(t, (1, 0, -0.5 * out_w, 0, 1, -0.5 * out_h))
    if "angle" in coeff:
        a = float(coeff["angle"])
        t = _left_multiply(t, (np.cos(a), -np.sin(a), 0, np.sin(a), np.cos(a), 0))
    if "dx" in coeff or "dy" in coeff:
        t = _left_multiply(t, (1, 0, coeff.get("dx", 0.0) * out_w, 0, 1, coeff.get("dy", 0.0) * out_h))
    if "zoom_x" in coeff or "zoom_y" in coeff:
        t = _left_multiply(t, (1.0 / coeff.get("zoom_x", 1.0), 0, 0, 0, 1.0 / coeff.get("zoom_y", 1.0), 0))
    t = _left_multiply(t, (1, 0, 0.5 * src_w, 0, 1, 0.5 * src_h))
    return np.array(t, F32)


def transmat_inverse(t):
    a, b, c, d, e, f = (F32(v) for v in t)
    den = a * e - b * d
    return np.array([e / den, b / -den, (c * e - b * f) / -den, d / -den, a / den, (c * d - a * f) / den], F32)


def corners_fit(coeff, src_w, src_h, out_w, out_h):
    """True when the four corners of the output land inside [0, size-2] of the source."""
    ang = coeff.get("angle", 0.0)
    for x in (0, out_w - 1):
        for y in (0, out_h - 1):
            x1, y1 = x - 0.5 * out_w, y - 0.5 * out_h
            x2 = np.cos(ang) * x1 - np.sin(ang) * y1 + coeff.get("dx", 0.0) * out_w
            y2 = np.sin(ang) * x1 + np.cos(ang) * y1 + coeff.get("dy", 0.0) * out_h
            x2 = x2 / coeff.get("zoom_x", 1.0) + 0.5 * src_w
            y2 = y2 / coeff.get("zoom_y", 1.0) + 0.5 * src_h
            if np.floor(x2) < 0 or np.floor(x2) > src_w - 2.0 or np.floor(y2) < 0 or np.floor(y2) > src_h - 2.0:
                return False
    return True


def augment(src, trans, chroma, out_h, out_w):
    src = np.asarray(src, F32)
    N, SH, SW, C = src.shape
    y, x = np.meshgrid(np.arange(out_h, dtype=F32), np.arange(out_w, dtype=F32), indexing="ij")
    out = np.empty((N, out_h, out_w, C), F32)
    for n in range(N):
        t = np.asarray(trans[n], F32)
        xp = np.clip(x * t[0] + y * t[1] + t[2], F32(0), F32(SW) - F32(1.05)).astype(F32)
        yp = np.clip(x * t[3] + y * t[4] + t[5], F32(0), F32(SH) - F32(1.05)).astype(F32)
        tlx, tly = np.floor(xp), np.floor(yp)
        xd, yd = (xp - tlx)[..., None], (yp - tly)[..., None]
        ix, iy = tlx.astype(int), tly.astype(int)
        img = src[n]
        dest = ((1 - xd) * (1 - yd) * img[iy, ix] + xd * yd * img[iy + 1, ix + 1] + (1 - xd) * yd * img[iy + 1, ix]
                + xd * (1 - yd) * img[iy, ix + 1]).astype(F32)
        if chroma is not None:
            g, b, c = (F32(v) for v in chroma[n][:3])
            rgb = dest * np.asarray(chroma[n][3:6], F32)
            comp = dest.sum(-1, dtype=F32) / (rgb.sum(-1, dtype=F32) + F32(0.01))
            v = np.clip(rgb * comp[..., None], 0, 1).astype(F32)
            v = np.power(v, g, dtype=F32) + b
            dest = np.clip(F32(0.5) + (v - F32(0.5)) * c, 0, 1).astype(F32)
        out[n] = dest
    return out


def augment_loops(src, trans, chroma, out_h, out_w):
    src = np.asarray(src, F32)
    N, SH, SW, C = src.shape
    out = np.empty((N, out_h, out_w, C), F32)
    for n in range(N):
        t = [F32(v) for v in trans[n]]
        for y in range(out_h):
            for x in range(out_w):
                xp = min(max(F32(x) * t[0] + F32(y) * t[1] + t[2], F32(0)), F32(SW) - F32(1.05))
                yp = min(max(F32(x) * t[3] + F32(y) * t[4] + t[5], F32(0)), F32(SH) - F32(1.05))
                tlx, tly = int(np.floor(xp)), int(np.floor(yp))
                xd, yd = F32(xp - tlx), F32(yp - tly)
                rgb, mean_in, mean_out = [], F32(0), F32(0)
                for c in range(C):
                    d = ((1 - xd) * (1 - yd) * src[n, tly, tlx, c] + xd * yd * src[n, tly + 1, tlx + 1, c]
                         + (1 - xd) * yd * src[n, tly + 1, tlx, c] + xd * (1 - yd) * src[n, tly, tlx + 1, c])
                    if chroma is None:
                        out[n, y, x, c] = d
                    else:
                        mean_in += d
                        rgb.append(F32(d) * F32(chroma[n][3 + c]))
                        mean_out += rgb[-1]
                if chroma is not None:
                    comp = mean_in / (mean_out + F32(0.01))
                    for c in range(C):
                        v = min(max(rgb[c] * comp, F32(0)), F32(1))
                        v = F32(np.power(F32(v), F32(chroma[n][0]))) + F32(chroma[n][1])
                        v = F32(0.5) + (v - F32(0.5)) * F32(chroma[n][2])
                        out[n, y, x, c] = min(max(v, F32(0)), F32(1))
    return out


def flow_augmentation(flows, trans_a, inv_trans_b, out_h, out_w):
    flows = np.asarray(flows, F32)
    N, SH, SW, _ = flows.shape
    flat = flows.reshape(-1)
    y, x = np.meshgrid(np.arange(out_h, dtype=F32), np.arange(out_w, dtype=F32), indexing="ij")
    out = np.empty((N, out_h, out_w, 2), F32)
    for n in range(N):
        a, b = np.asarray(trans_a[n], F32), np.asarray(inv_trans_b[n], F32)
        x1 = x * a[0] + y * a[1] + a[2]
        y1 = x * a[3] + y * a[4] + a[5]
        idx = ((n * SH + (y1 + F32(0.5)).astype(np.int64)) * SW + (x1 + F32(0.5)).astype(np.int64)) * 2
        x2 = x1 + flat[np.clip(idx, 0, flat.size - 1)]
        y2 = y1 + flat[np.clip(idx + 1, 0, flat.size - 1)]
        out[n, ..., 0] = x2 * b[0] + y2 * b[1] + b[2] - x
        out[n, ..., 1] = x2 * b[3] + y2 * b[4] + b[5] - y
    return out


def flow_augmentation_loops(flows, trans_a, inv_trans_b, out_h, out_w):
    flows = np.asarray(flows, F32)
    N, SH, SW, _ = flows.shape
    flat = flows.reshape(-1)
    out = np.empty((N, out_h, out_w, 2), F32)
    for n in range(N):
        a, b = [F32(v) for v in trans_a[n]], [F32(v) for v in inv_trans_b[n]]
        for y in range(out_h):
            for x in range(out_w):
                x1 = F32(x) * a[0] + F32(y) * a[1] + a[2]
                y1 = F32(x) * a[3] + F32(y) * a[4] + a[5]
                ix = ((n * SH + int(y1 + F32(0.5))) * SW + int(x1 + F32(0.5))) * 2
                x2 = x1 + flat[min(max(ix, 0), flat.size - 1)]
                y2 = y1 + flat[min(max(ix + 1, 0), flat.size - 1)]
                out[n, y, x, 0] = x2 * b[0] + y2 * b[1] + b[2] - F32(x)
                out[n, y, x, 1] = x2 * b[3] + y2 * b[4] + b[5] - F32(y)
    return out
