"""Oracle (test infrastructure): the TensorFlow-builtin arithmetic the model
files call, restated in NumPy.

PARITY UNPINNED: TensorFlow 1.x / cuDNN are third-party, not vendored and not
installed here; the reference has no tests at this boundary.  What is restated
is the published definition of each TF op at the reference's call sites
(src/utils.py:401-421, src/flownet_s/flownet_s.py:26-111).  tests/test_oracle_nn.py
cross-checks every function against torch's CPU kernels.

All tensors NHWC.  Computation in float64 by default (``dtype``), so that the
oracle is the exact-arithmetic meaning of the reference graph and both the
fp32 and the bf16 GPU paths can be measured against it.
"""
import numpy as np


def pad(x, num=1):
    """utils.py:408-412 -- zero pad H and W by ``num`` on both sides."""
    return np.pad(x, ((0, 0), (num, num), (num, num), (0, 0)))


def antipad(x, num=1):
    """utils.py:415-421 -- crop ``num`` pixels on each side of H and W."""
    return x[:, num:x.shape[1] - num, num:x.shape[2] - num, :]


def leaky_relu(x, leak=0.1):
    """utils.py:401-405 -- 0.5(1+leak) x + 0.5(1-leak)|x|."""
    f1 = 0.5 * (1.0 + leak)
    f2 = 0.5 * (1.0 - leak)
    return f1 * x + f2 * np.abs(x)


def conv2d(x, w, b=None, stride=1, padding=0, activation=None, dtype=np.float64):
    """slim.conv2d(pad(x, padding), Cout, k, stride, padding='VALID')
    (flownet_s.py:39-50): out[n,y,x,o] = act(b[o] + sum_{ky,kx,i}
    xpad[n, y*s+ky, x*s+kx, i] * w[ky,kx,i,o]); w is HWIO; no kernel flip."""
    x = np.asarray(x, dtype)
    w = np.asarray(w, dtype)
    if padding:
        x = pad(x, padding)
    N, H, W, Cin = x.shape
    kh, kw, wcin, Cout = w.shape
    assert wcin == Cin, (wcin, Cin)
    oh = (H - kh) // stride + 1
    ow = (W - kw) // stride + 1
    sN, sH, sW, sC = x.strides
    cols = np.lib.stride_tricks.as_strided(
        x, shape=(N, oh, ow, kh, kw, Cin),
        strides=(sN, sH * stride, sW * stride, sH, sW, sC), writeable=False)
    out = cols.reshape(N * oh * ow, kh * kw * Cin) @ w.reshape(kh * kw * Cin, Cout)
    out = out.reshape(N, oh, ow, Cout)
    if b is not None:
        out = out + np.asarray(b, dtype)
    if activation is not None:
        out = activation(out)
    return out


def conv2d_transpose(x, w, stride=2, crop=1, activation=None, dtype=np.float64, bias=None):
    """antipad(slim.conv2d_transpose(x, Cout, 4, stride=2, padding='VALID'))
    (flownet_s.py:53-63): full[n, s*y+ky, s*x+kx, o] += x[n,y,x,i] * w[ky,kx,o,i]
    (w is HW-O-I), size s(H-1)+k; + bias where the layer has one (slim adds it before
    the activation; none under biases_initializer=None, flownet_s.py:53 -- the FlowNet2
    fusion net's transposed convs have one, flownet2.py:50-89); then crop."""
    x = np.asarray(x, dtype)
    w = np.asarray(w, dtype)
    N, H, W, Cin = x.shape
    kh, kw, Cout, wcin = w.shape
    assert wcin == Cin, (wcin, Cin)
    full = np.zeros((N, stride * (H - 1) + kh, stride * (W - 1) + kw, Cout), dtype)
    for ky in range(kh):
        for kx in range(kw):
            full[:, ky:ky + stride * (H - 1) + 1:stride, kx:kx + stride * (W - 1) + 1:stride, :] += \
                x @ w[ky, kx].T
    if bias is not None:
        full = full + np.asarray(bias, dtype)
    out = antipad(full, crop) if crop else full
    if activation is not None:
        out = activation(out)
    return out


def resize_bilinear_align_corners(x, size, dtype=np.float64):
    """tf.image.resize_bilinear(x, size, align_corners=True) (flownet_s.py:109-111):
    scale=(in-1)/(out-1); src=dst*scale; lo=floor(src); hi=min(lo+1,in-1)."""
    x = np.asarray(x, dtype)
    N, H, W, C = x.shape
    oh, ow = int(size[0]), int(size[1])

    def axis(n_in, n_out):
        scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
        src = np.arange(n_out, dtype=np.float64) * scale
        lo = np.floor(src).astype(np.int64)
        hi = np.minimum(lo + 1, n_in - 1)
        return lo, hi, (src - lo)

    ylo, yhi, fy = axis(H, oh)
    xlo, xhi, fx = axis(W, ow)
    fy = fy[None, :, None, None]
    fx = fx[None, None, :, None]
    top = x[:, ylo][:, :, xlo] * (1 - fx) + x[:, ylo][:, :, xhi] * fx
    bot = x[:, yhi][:, :, xlo] * (1 - fx) + x[:, yhi][:, :, xhi] * fx
    return (top * (1 - fy) + bot * fy).astype(dtype)


def channel_norm(x):
    """flownet2.py:25-28 / flownet_cs.py:24-27 -- sqrt(sum_c x^2), keepdims."""
    return np.sqrt(np.sum(np.square(x), axis=3, keepdims=True))
