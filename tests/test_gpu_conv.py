"""fn2_conv2d (implicit GEMM on MFMA) against the oracle's conv2d / conv2d_transpose,
fp32 path (parity bar) and bf16 path (bounded by bf16 input rounding)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import nn as refnn

pytestmark = pytest.mark.gpu


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def run_conv(x, w, b, kind, k, stride, pad, act, dtype, out_f32=False, cin_off=0, cout_off=0, extra_out=0,
             use_ws=True, force_generic=False):
    """Drive the C ABI directly.  x: [N,H,W,Cin] fp32.  Returns fp32 numpy [N,oh,ow,Cout]."""
    from src import _hip, weights as W
    lib = _hip.lib()
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    N, H, Wd, cin = x.shape
    cout = w.shape[3] if kind == "conv" else w.shape[2]
    cs_in = (cin_off + cin + 63) // 64 * 64 if cin > 32 else (cin_off + cin + 7) // 8 * 8
    xin = torch.zeros((N, H, Wd, cs_in), dtype=td, device="cuda")
    xin[..., cin_off:cin_off + cin] = torch.from_numpy(x).cuda().to(td)
    if kind == "conv":
        oh, ow = (H + 2 * pad - k) // stride + 1, (Wd + 2 * pad - k) // stride + 1
    else:
        oh, ow = 2 * H, 2 * Wd
    od = torch.float32 if (out_f32 or dtype == "f32") else torch.bfloat16
    cs_out = cout_off + cout + extra_out
    out = torch.full((N, oh, ow, cs_out), 7.0, dtype=od, device="cuda")
    tile = lib.fn2_conv2d_cout_tile(cout)
    kstep = 32 if dtype == "bf16" else 16
    code = 1 if dtype == "bf16" else 0
    cin_pad = (cin + 7) // 8 * 8
    cin64 = (cin + 63) // 64 * 64
    if not force_generic and cin > 32 and lib.fn2_conv2d_weight_layout(code, cin64, cout) == 1:
        cin_pad = cin64
    layout = 0 if force_generic else lib.fn2_conv2d_weight_layout(code, cin_pad, cout)
    run_conv.last_layout = layout
    if kind == "conv":
        packed, cin_pad, cout_pad, kpad = W.pack_conv(w, tile, kstep, cin_pad, layout)
    else:
        packed, cin_pad, cout_pad, kpad = W.pack_deconv(w, tile, kstep, cin_pad, layout)
    wdev = torch.from_numpy(packed).cuda().to(td).contiguous()
    bdev = torch.from_numpy(b).cuda() if b is not None else None
    d = _hip.Fn2ConvDesc()
    d.inp = _hip.view(xin, cin, cin_off)
    d.out = _hip.view(out, cout, cout_off)
    d.wgt = wdev.data_ptr()
    d.bias = bdev.data_ptr() if bdev is not None else None
    d.kind = 0 if kind == "conv" else 1
    d.kh = d.kw = k
    d.stride, d.pad = stride, pad
    d.act = 1 if act else 0
    d.cin_pad, d.cout_pad, d.kpad = cin_pad, cout_pad, kpad
    d.wgt_layout = layout
    need = int(lib.fn2_conv2d_workspace_bytes(C.byref(d)))
    ws = None
    if use_ws and need > 0:
        ws = torch.full(((need + 3) // 4,), float("nan"), dtype=torch.float32, device="cuda")  # poisoned
        d.workspace, d.workspace_bytes = ws.data_ptr(), need
    run_conv.last_ws_bytes = need if use_ws else 0
    _hip.check(lib.fn2_conv2d(C.byref(d), _hip.stream_ptr()))
    torch.cuda.synchronize()
    res = out.float().cpu().numpy()
    # nothing outside the slice may be touched
    if cout_off:
        assert np.all(res[..., :cout_off] == 7.0)
    if extra_out:
        assert np.all(res[..., cout_off + cout:] == 7.0)
    return res[..., cout_off:cout_off + cout]


CONV_CASES = [
    # k, stride, pad, cin, cout, H, W      (the layer shapes of flownet_s.py:39-50 at reduced size)
    (7, 2, 3, 6, 64, 32, 48),      # conv1 stem: Cin 6 -> 8 channel padding
    (7, 2, 3, 3, 64, 32, 32),      # FlowNetC conv1: Cin 3
    (5, 2, 2, 64, 128, 24, 32),    # conv2
    (3, 1, 1, 256, 256, 6, 8),     # conv3_1
    (3, 2, 1, 128, 512, 12, 16),   # stride-2 3x3, 128-cout tile x4
    (3, 1, 1, 194, 2, 12, 16),     # predict_flow2: Cin 194 (pad 200), Cout 2 -> 16-cout tile, scalar stores
    (1, 1, 0, 256, 32, 6, 8),      # conv_redir 1x1
    (3, 1, 1, 11, 64, 16, 16),     # fuse_conv0: Cin 11
    (3, 1, 1, 82, 16, 16, 16),     # fuse_interconv0
    (3, 1, 1, 473, 256, 6, 8),     # FlowNetC conv3_1: Cin 473
]


@pytest.mark.parametrize("k,s,p,cin,cout,H,W", CONV_CASES)
def test_conv_f32_matches_oracle(k, s, p, cin, cout, H, W):
    x = rnd((2, H, W, cin), 0)
    w = rnd((k, k, cin, cout), 1, (2.0 / (k * k * cin)) ** 0.5)
    b = rnd((cout,), 2, 0.1)
    want = refnn.conv2d(x, w, b, stride=s, padding=p, activation=refnn.leaky_relu)
    got = run_conv(x, w, b, "conv", k, s, p, True, "f32")
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)  # fp32 accumulation-order tolerance


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fast_and_generic_kernels_agree(dtype):
    """The LDS-DMA kernel (permuted weight rows) and the generic kernel on the same layer."""
    x = torch.from_numpy(rnd((2, 12, 16, 128), 20)).bfloat16().float().numpy()
    w = torch.from_numpy(rnd((3, 3, 128, 192), 21, (2.0 / (9 * 128)) ** 0.5)).bfloat16().float().numpy()
    b = rnd((192,), 22, 0.1)
    want = refnn.conv2d(x, w, b, stride=2, padding=1, activation=refnn.leaky_relu)
    fast = run_conv(x, w, b, "conv", 3, 2, 1, True, dtype, out_f32=True, cout_off=8, use_ws=False)
    assert run_conv.last_layout == 1
    gen = run_conv(x, w, b, "conv", 3, 2, 1, True, dtype, out_f32=True, cout_off=8, use_ws=False, force_generic=True)
    assert run_conv.last_layout == 0
    np.testing.assert_allclose(fast, want, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(gen, want, rtol=2e-5, atol=2e-5)


def test_conv_f32_linear_no_bias_and_slices():
    x = rnd((1, 10, 12, 24), 3)
    w = rnd((3, 3, 24, 40), 4, 0.1)
    want = refnn.conv2d(x, w, None, stride=1, padding=1)
    got = run_conv(x, w, None, "conv", 3, 1, 1, False, "f32", cin_off=8, cout_off=16, extra_out=4)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("cin,cout,H,W", [(1024, 512, 3, 4), (386, 64, 6, 8), (162, 16, 8, 8), (128, 32, 5, 7)])
def test_deconv_f32_matches_oracle(cin, cout, H, W):
    x = rnd((2, H, W, cin), 5)
    w = rnd((4, 4, cout, cin), 6, (2.0 / (4 * cin)) ** 0.5)
    want = refnn.conv2d_transpose(x, w, activation=refnn.leaky_relu)
    got = run_conv(x, w, None, "deconv", 4, 2, 1, True, "f32", cout_off=8)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("k,s,p,cin,cout,H,W", CONV_CASES[:6])
def test_conv_bf16_matches_oracle_on_bf16_rounded_inputs(k, s, p, cin, cout, H, W):
    """bf16 path: identical to the fp64 oracle fed the bf16-rounded inputs/weights, up to fp32
    accumulation order; output kept in fp32 so that only the input rounding is modelled."""
    x = torch.from_numpy(rnd((2, H, W, cin), 0)).bfloat16().float().numpy()
    w = torch.from_numpy(rnd((k, k, cin, cout), 1, (2.0 / (k * k * cin)) ** 0.5)).bfloat16().float().numpy()
    b = rnd((cout,), 2, 0.1)
    want = refnn.conv2d(x, w, b, stride=s, padding=p, activation=refnn.leaky_relu)
    got = run_conv(x, w, b, "conv", k, s, p, True, "bf16", out_f32=True)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)
    got16 = run_conv(x, w, b, "conv", k, s, p, True, "bf16")
    np.testing.assert_allclose(got16, want, rtol=1e-2, atol=1e-2)  # + one bf16 output rounding


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_splitk_equals_single_pass(dtype):
    """Small output grid + long K (conv6_1-like): the split-K path (workspace given) and the
    single-pass path (no workspace) agree with the oracle and with each other."""
    x = torch.from_numpy(rnd((2, 6, 8, 256), 9)).bfloat16().float().numpy()
    w = torch.from_numpy(rnd((3, 3, 256, 192), 10, (2.0 / (9 * 256)) ** 0.5)).bfloat16().float().numpy()
    b = rnd((192,), 11, 0.1)
    want = refnn.conv2d(x, w, b, stride=1, padding=1, activation=refnn.leaky_relu)
    got_split = run_conv(x, w, b, "conv", 3, 1, 1, True, dtype, out_f32=True, cout_off=8)
    assert run_conv.last_ws_bytes > 0  # this shape must take the split-K path
    got_single = run_conv(x, w, b, "conv", 3, 1, 1, True, dtype, out_f32=True, cout_off=8, use_ws=False)
    np.testing.assert_allclose(got_split, want, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(got_single, want, rtol=2e-5, atol=2e-5)
    # deconv phases with split-K
    xd = rnd((1, 3, 4, 512), 12)
    wd = rnd((4, 4, 256, 512), 13, (2.0 / (4 * 512)) ** 0.5)
    wantd = refnn.conv2d_transpose(xd, wd, activation=refnn.leaky_relu)
    gotd = run_conv(xd, wd, None, "deconv", 4, 2, 1, True, "f32")
    assert run_conv.last_ws_bytes > 0
    np.testing.assert_allclose(gotd, wantd, rtol=2e-5, atol=2e-5)


def test_upsample_flow_matches_oracle():
    from src import _hip
    lib = _hip.lib()
    x = rnd((2, 5, 7, 2), 7)
    w = rnd((4, 4, 2, 2), 8, 0.5)
    want = refnn.conv2d_transpose(x, w)
    out = torch.full((2, 10, 14, 8), 3.0, device="cuda")
    v = _hip.view(out, 2, 4)
    xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()  # keep alive across the async launch
    _hip.check(lib.fn2_upsample_flow(_hip.ptr(xd), _hip.ptr(wd), C.byref(v), 2, 5, 7, _hip.stream_ptr()))
    res = out.cpu().numpy()
    np.testing.assert_allclose(res[..., 4:6], want, rtol=1e-5, atol=1e-5)
    assert np.all(res[..., :4] == 3.0) and np.all(res[..., 6:] == 3.0)


def test_conv_rejects_bad_descriptors():
    from src import _hip
    lib = _hip.lib()
    d = _hip.Fn2ConvDesc()
    assert lib.fn2_conv2d(C.byref(d), None) == _hip.ERR_INVALID_ARGUMENT
    with pytest.raises(ValueError):
        _hip.check(lib.fn2_conv2d(C.byref(d), None))
