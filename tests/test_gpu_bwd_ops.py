"""The filter-gradient launch (fn2_conv2d_bwd_filter) and the flow-head tensor G18 (fn2_head_g18) on the C ABI against
plain NumPy sums: both pixel walks of the split-fp16 kernel (uniform: widths that divide / are divided by 32; general:
any width), the fused bias gradient, stride-2 and transposed layers, and the flow head's kind-4 form."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def to_dev(x, dtype):
    from src import weights as W
    if dtype == "f16x2":
        return torch.from_numpy(W.split_f16x2(x).view(np.float32)).cuda()
    return torch.from_numpy(x).cuda()


def filter_grad(x, dy, k, stride, pad):
    """dW[ky, kx, ci, co] = sum_{n,y,x} xpad[n, y s + ky, x s + kx, ci] dy[n, y, x, co]  (float64)."""
    n, h, w, ci = x.shape
    _, oh, ow, co = dy.shape
    xp = np.zeros((n, h + 2 * pad, w + 2 * pad, ci), np.float64)
    xp[:, pad:pad + h, pad:pad + w] = x
    out = np.zeros((k, k, ci, co), np.float64)
    d = dy.astype(np.float64).reshape(-1, co)
    for ky in range(k):
        for kx in range(k):
            win = xp[:, ky:ky + (oh - 1) * stride + 1:stride, kx:kx + (ow - 1) * stride + 1:stride]
            out[ky, kx] = win.reshape(-1, ci).T @ d
    return out


@pytest.mark.parametrize("dtype", ["f16x2", "f32"])
@pytest.mark.parametrize("k,stride,pad,cin,cout,H,W", [
    (3, 1, 1, 64, 96, 8, 64),      # uniform walk: 32 pixels of one row per stage
    (3, 1, 1, 136, 64, 12, 16),    # uniform walk: two whole rows per stage; ragged channels
    (3, 2, 1, 64, 128, 24, 48),    # general walk (dy is 12 x 24 wide): stride 2
    (5, 2, 2, 32, 64, 16, 40),     # general walk, 5x5
    (3, 1, 1, 64, 64, 6, 8),       # general walk: stages straddle images (6 rows of 8)
])
def test_filter_gradient_matches_numpy(dtype, k, stride, pad, cin, cout, H, W):
    from src import _hip
    lib = _hip.lib()
    N = 3
    oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x, dy = rnd((N, H, W, cin), 70), rnd((N, oh, ow, cout), 71)
    want = filter_grad(x, dy, k, stride, pad)                     # [k, k, ci, co]
    cs_x, cs_y = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
    xp = np.zeros((N, H, W, cs_x), np.float32); xp[..., :cin] = x
    yp = np.zeros((N, oh, ow, cs_y), np.float32); yp[..., :cout] = dy
    xd, yd = to_dev(xp, dtype), to_dev(yp, dtype)
    code = 3 if dtype == "f16x2" else 0
    cin_pad, kpad = cs_x, k * k * cs_x
    dw = torch.zeros(cout * kpad, dtype=torch.float32, device="cuda")
    db = torch.zeros(cout, dtype=torch.float32, device="cuda")
    d = _hip.Fn2BwdwDesc()
    d.x, d.dy, d.dw, d.db = _hip.view(xd, cin, 0, code), _hip.view(yd, cout, 0, code), dw.data_ptr(), db.data_ptr()
    d.kind, d.kh, d.kw, d.stride, d.pad = 0, k, k, stride, pad
    d.cin_pad, d.cout_pad, d.kpad, d.wgt_layout = cin_pad, cout, kpad, 0   # natural rows: dw[co][tap * cin_pad + ci]
    _hip.check(lib.fn2_conv2d_bwd_filter(C.byref(d), _hip.stream_ptr()))
    torch.cuda.synchronize()
    got = dw.cpu().numpy().reshape(cout, k, k, cin_pad)[..., :cin].transpose(1, 2, 3, 0)
    tol = 3e-6 * np.abs(want).max() * (4 if dtype == "f16x2" else 1)
    assert np.abs(got - want).max() < tol, (np.abs(got - want).max(), tol)
    want_b = dy.astype(np.float64).sum((0, 1, 2))
    assert np.abs(db.cpu().numpy() - want_b).max() < 1e-5 * np.abs(want_b).max() + 1e-4


def test_flow_head_gradients_through_g18():
    """fn2_head_g18 + kind 4: the head's filter gradient equals the plain sums, in the head's natural weight layout."""
    from src import _hip
    lib = _hip.lib()
    N, H, W, cin = 2, 12, 16, 72
    x, g = rnd((N, H, W, cin), 80), rnd((N, H, W, 2), 81)
    want = filter_grad(x, g, 3, 1, 1)                              # [3, 3, ci, 2]
    xd = to_dev(x, "f16x2")
    gd = torch.from_numpy(g).cuda()
    g18 = torch.zeros((N, H, W, 32), dtype=torch.float32, device="cuda")
    v18 = _hip.view(g18, 18, 0, 3)
    _hip.check(lib.fn2_head_g18(_hip.ptr(gd), C.byref(v18), _hip.stream_ptr()))
    from src import weights as Wt
    t = Wt.join_f16x2(g18.cpu().numpy().view(np.float16))          # [N, H, W, 32]
    gp = np.zeros((N, H + 2, W + 2, 2), np.float32); gp[:, 1:-1, 1:-1] = g
    for tap in range(9):
        ky, kx = divmod(tap, 3)
        np.testing.assert_allclose(t[..., tap * 2:tap * 2 + 2], gp[:, 2 - ky:2 - ky + H, 2 - kx:2 - kx + W], rtol=0, atol=1e-6)
    assert np.all(t[..., 18:] == 0)
    cin_pad, kpad = 72, 9 * 72
    dw = torch.zeros(2 * kpad, dtype=torch.float32, device="cuda")
    d = _hip.Fn2BwdwDesc()
    d.x, d.dy, d.dw = _hip.view(xd, cin, 0, 3), v18, dw.data_ptr()
    d.kind, d.kh, d.kw, d.stride, d.pad = 4, 3, 3, 1, 1
    d.cin_pad, d.cout_pad, d.kpad, d.wgt_layout = cin_pad, 32, kpad, 0
    _hip.check(lib.fn2_conv2d_bwd_filter(C.byref(d), _hip.stream_ptr()))
    torch.cuda.synchronize()
    got = dw.cpu().numpy().reshape(2, 3, 3, cin_pad)[..., :cin].transpose(1, 2, 3, 0)
    assert np.abs(got - want).max() < 1.2e-5 * np.abs(want).max()


@pytest.mark.parametrize("dtype", ["f16x2", "f32"])
@pytest.mark.parametrize("k,pad,co,ci,H,W", [(3, 1, 128, 64, 12, 16), (5, 2, 64, 96, 8, 16), (3, 1, 256, 256, 6, 8)])
def test_input_gradient_of_stride2_conv(dtype, k, pad, co, ci, H, W):
    """fn2_conv2d kind 3 (the transpose of a k x k stride-2 convolution: four phases that walk only their non-zero tap
    slots) against the plain scatter dx[n, 2y + ky - p, 2x + kx - p, ci] += dy[n, y, x, co] w[ky, kx, ci, co]."""
    from src import _hip, weights as Wt
    lib = _hip.lib()
    N = 2
    dy = rnd((N, H, W, co), 90)
    w = rnd((k, k, ci, co), 91, (2.0 / (k * k * co)) ** 0.5)
    want = np.zeros((N, 2 * H + k, 2 * W + k, ci), np.float64)   # scatter into a padded frame, then crop
    for ky in range(k):
        for kx in range(k):
            want[:, ky:ky + 2 * H:2, kx:kx + 2 * W:2] += dy.astype(np.float64) @ w[ky, kx].astype(np.float64).T
    want = want[:, pad:pad + 2 * H, pad:pad + 2 * W]
    code = 3 if dtype == "f16x2" else 0
    esz_line = 32
    cin_pad = (co + esz_line - 1) // esz_line * esz_line
    plan = _hip.conv_plan(code, cin_pad, ci)
    if plan.layout != 1:
        cin_pad = (co + 7) // 8 * 8
        plan = _hip.conv_plan(code, cin_pad, ci)
    packed, cin_pad, cout_pad, kpad = Wt.pack_conv_transpose_s2(w, pad, plan.cout_tile, plan.kstep_elems, cin_pad, plan.layout)
    scale = 1.0
    if plan.wgt_dtype == 3:
        k2 = int(np.floor(np.log2(1024.0 / np.abs(packed).max())))
        packed, scale = packed * 2.0 ** k2, 2.0 ** -k2
    wdev = Wt.packed_to_device(packed, plan.wgt_dtype, "cuda")
    yp = np.zeros((N, H, W, cin_pad), np.float32); yp[..., :co] = dy
    yd = to_dev(yp, dtype)
    cs_out = (ci + 7) // 8 * 8
    out = to_dev(np.zeros((N, 2 * H, 2 * W, cs_out), np.float32), dtype)
    d = _hip.Fn2ConvDesc()
    d.inp, d.out = _hip.view(yd, co, 0, code), _hip.view(out, ci, 0, code)
    d.wgt, d.bias = wdev.data_ptr(), None
    d.kind, d.kh, d.kw, d.stride, d.pad, d.act = 3, k, k, 2, pad, 0
    d.cin_pad, d.cout_pad, d.kpad, d.wgt_layout, d.out_scale = cin_pad, cout_pad, kpad, plan.layout, scale
    need = int(lib.fn2_conv2d_workspace_bytes(C.byref(d)))
    if need:
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device="cuda")
        d.workspace, d.workspace_bytes = ws.data_ptr(), need
    _hip.check(lib.fn2_conv2d(C.byref(d), _hip.stream_ptr()))
    torch.cuda.synchronize()
    from src import weights as W2
    got = W2.join_f16x2(out.cpu().numpy().view(np.float16)) if code == 3 else out.cpu().numpy()
    np.testing.assert_allclose(got[..., :ci], want, rtol=2e-5, atol=2e-5)
