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


_TD = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16, "f16x2": torch.float32}
_CODE = {"f32": 0, "bf16": 1, "f16": 2, "f16x2": 3}


def _to_dev(x, dtype):
    """fp32 NHWC numpy -> device tensor in the engine's storage format."""
    from src import weights as W
    if dtype == "f16x2":
        return torch.from_numpy(W.split_f16x2(x).view(np.float32)).cuda()
    return torch.from_numpy(x).cuda().to(_TD[dtype])


def _from_dev(t, code):
    from src import weights as W
    if code == 3:
        return W.join_f16x2(t.cpu().numpy().view(np.float16))
    return t.float().cpu().numpy()


def run_conv(x, w, b, kind, k, stride, pad, act, dtype, out_f32=False, cin_off=0, cout_off=0, extra_out=0,
             use_ws=True, force_generic=False, in_f32=False, act_grad=None, fragments=False):
    """Drive the C ABI directly.  x: [N,H,W,Cin] fp32.  Returns fp32 numpy [N,oh,ow,Cout].
    in_f32: fp32 input into a non-fp32 engine format (the stem of a split-fp16 network)."""
    from src import _hip, weights as W
    lib = _hip.lib()
    N, H, Wd, cin = x.shape
    cout = w.shape[3] if kind == "conv" else w.shape[2]
    in_code = 0 if in_f32 else _CODE[dtype]
    cs_in = (cin_off + cin + 63) // 64 * 64 if cin > 32 else (cin_off + cin + 7) // 8 * 8
    xpad = np.zeros((N, H, Wd, cs_in), np.float32)
    xpad[..., cin_off:cin_off + cin] = x
    xin = _to_dev(xpad, "f32" if in_f32 else dtype)
    if kind == "conv":
        oh, ow = (H + 2 * pad - k) // stride + 1, (Wd + 2 * pad - k) // stride + 1
    else:
        oh, ow = 2 * H, 2 * Wd
    out_code = 0 if (out_f32 or dtype == "f32") else _CODE[dtype]
    cs_out = cout_off + cout + extra_out
    if out_code == 3:
        cs_out = (cs_out + 7) // 8 * 8
    out = _to_dev(np.full((N, oh, ow, cs_out), 7.0, np.float32), "f32" if out_code == 0 else dtype)
    esz = 2 if in_code in (1, 2) else 4
    cin_pad = (cin + 7) // 8 * 8
    cin_line = (cin + 128 // esz - 1) // (128 // esz) * (128 // esz)
    if not force_generic and cin_off + cin_line <= cs_in and _hip.conv_plan(in_code, cin_line, cout).layout == 1:
        cin_pad = cin_line
    plan = _hip.conv_plan(in_code, cin_pad, cout)
    run_conv.last_layout = plan.layout
    pack = W.pack_conv if kind == "conv" else W.pack_deconv
    packed, cin_pad, cout_pad, kpad = pack(w, plan.cout_tile, plan.kstep_elems, cin_pad, plan.layout)
    out_scale = 1.0
    if plan.wgt_dtype == 3:
        k2 = int(np.floor(np.log2(1024.0 / np.abs(packed).max())))
        packed, out_scale = packed * 2.0 ** k2, 2.0 ** -k2
    wdev = W.packed_to_device(packed, plan.wgt_dtype, "cuda")
    if fragments:  # wgt_layout 2: the packed matrix re-tiled into MFMA-fragment order (weights straight to registers)
        wdev = W.to_fragment_order(wdev)
    bdev = torch.from_numpy(b).cuda() if b is not None else None
    d = _hip.Fn2ConvDesc()
    d.inp = _hip.view(xin, cin, cin_off, in_code)
    d.out = _hip.view(out, cout, cout_off, out_code)
    d.wgt = wdev.data_ptr()
    d.bias = bdev.data_ptr() if bdev is not None else None
    d.kind = 0 if kind == "conv" else 1
    d.kh = d.kw = k
    d.stride, d.pad = stride, pad
    d.act = 1 if act else 0
    d.cin_pad, d.cout_pad, d.kpad = cin_pad, cout_pad, kpad
    d.wgt_layout = 2 if fragments else plan.layout
    d.out_scale = out_scale
    if act_grad is not None:  # (y [N,oh,ow,Cout] fp32, c0, c1): out += result, then the LeakyReLU factor from y on [c0, c1)
        ynp, d.act_grad_c0, d.act_grad_c1 = act_grad
        ybuf = np.full((N, oh, ow, cs_out), -3.0, np.float32)
        ybuf[..., cout_off:cout_off + cout] = ynp
        ydev = _to_dev(ybuf, "f32" if out_code == 0 else dtype)
        d.act_grad_y, d.accumulate = ydev.data_ptr(), 1
    need = int(lib.fn2_conv2d_workspace_bytes(C.byref(d)))
    ws = None
    if use_ws and need > 0:
        ws = torch.full(((need + 3) // 4,), float("nan"), dtype=torch.float32, device="cuda")  # poisoned
        d.workspace, d.workspace_bytes = ws.data_ptr(), need
    run_conv.last_ws_bytes = need if use_ws else 0
    _hip.check(lib.fn2_conv2d(C.byref(d), _hip.stream_ptr()))
    torch.cuda.synchronize()
    res = _from_dev(out, out_code)
    # nothing outside the slice may be touched
    if cout_off:
        assert np.all(res[..., :cout_off] == 7.0)
    if extra_out:
        assert np.all(res[..., cout_off + cout:cout_off + cout + extra_out] == 7.0)
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


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_fast_and_generic_kernels(dtype):
    """The same 3x3 stride-2 layer on the LDS-DMA kernel (Cin 128: whole 128-byte lines, permuted weight rows)
    and, with Cin 72, on the generic kernel."""
    b = rnd((192,), 22, 0.1)
    for cin, layout in ((128, 1), (72, 0)):
        x = torch.from_numpy(rnd((2, 12, 16, cin), 20)).bfloat16().float().numpy()
        w = torch.from_numpy(rnd((3, 3, cin, 192), 21, (2.0 / (9 * cin)) ** 0.5)).bfloat16().float().numpy()
        want = refnn.conv2d(x, w, b, stride=2, padding=1, activation=refnn.leaky_relu)
        got = run_conv(x, w, b, "conv", 3, 2, 1, True, dtype, out_f32=True, cout_off=8, use_ws=False,
                       force_generic=(layout == 0))
        assert run_conv.last_layout == layout
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)


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


@pytest.mark.parametrize("dtype", ["f32", "f16x2"])
@pytest.mark.parametrize("cin,cout,H,W", [(128, 32, 12, 16), (162, 16, 24, 32), (1024, 512, 3, 4), (72, 24, 5, 7)])
def test_deconv_with_bias_matches_oracle(cin, cout, H, W, dtype):
    """Transposed conv WITH a bias, added before the LeakyReLU as slim.conv2d_transpose does: the FlowNet2 fusion net's
    fuse_deconv1 (128 -> 32) and fuse_deconv0 (162 -> 16) (flownet2.py:66-84, no biases_initializer=None scope at
    :50-57); a split-K geometry (bias applied by the finalize pass) and an odd-sized one as well."""
    x = rnd((2, H, W, cin), 5)
    w = rnd((4, 4, cout, cin), 6, (2.0 / (4 * cin)) ** 0.5)
    b = rnd((cout,), 9, 0.5)
    want = refnn.conv2d_transpose(x, w, activation=refnn.leaky_relu, bias=b)
    assert np.abs(want - refnn.conv2d_transpose(x, w, activation=refnn.leaky_relu)).max() > 0.1  # the bias matters
    got = run_conv(x, w, b, "deconv", 4, 2, 1, True, dtype, cout_off=8, out_f32=True)
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


@pytest.mark.parametrize("dtype", ["f32", "f16x2"])
@pytest.mark.parametrize("kind,k,s,p,cin,cout,H,W,c0,c1", [
    ("conv", 3, 1, 1, 128, 128, 24, 32, 0, 128),     # whole view (the next encoder layer's input gradient); halo kernel
    ("conv", 3, 1, 1, 256, 194, 12, 16, 64, 128),    # a middle slice of a concat gradient
    ("conv", 3, 1, 1, 512, 512, 6, 8, 256, 512),     # split-K: the factor is applied by the finalize pass
    ("deconv", 4, 2, 1, 256, 96, 6, 8, 16, 96),      # four phases
    ("conv", 5, 1, 2, 64, 48, 16, 24, 16, 48),       # range ending at the view's ragged end
])
def test_fused_leaky_backward_in_epilogue(dtype, kind, k, s, p, cin, cout, H, W, c0, c1):
    """fn2_conv_desc.act_grad_y: out = (out + conv) * LeakyReLU'(y) on channels [c0, c1), untouched elsewhere."""
    x = rnd((2, H, W, cin), 60)
    if kind == "conv":
        w = rnd((k, k, cin, cout), 61, (2.0 / (k * k * cin)) ** 0.5)
        lin = refnn.conv2d(x, w, None, stride=s, padding=p)
    else:
        w = rnd((4, 4, cout, cin), 61, (2.0 / (4 * cin)) ** 0.5)
        lin = refnn.conv2d_transpose(x, w)
    y = rnd(lin.shape, 62)
    y[np.abs(y) < 0.2] = 0.0  # exact zeros take the factor 0.55 (tf.abs' = sign: utils.py:401-405)
    want = lin + 7.0
    f = np.where(y > 0, 1.0, np.where(y < 0, 0.1, 0.55)).astype(np.float32)
    want[..., c0:c1] *= f[..., c0:c1]
    got = run_conv(x, w, None, kind, k, s, p, False, dtype, cout_off=16, extra_out=16, act_grad=(y, c0, c1))
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)


X2_CASES = [(5, 2, 2, 64, 128, 24, 32), (3, 1, 1, 256, 256, 6, 8), (3, 2, 1, 128, 512, 12, 16),
            (3, 1, 1, 473, 256, 6, 8), (1, 1, 0, 256, 32, 6, 8), (3, 1, 1, 82, 16, 16, 16), (3, 1, 1, 162, 32, 8, 8)]


@pytest.mark.parametrize("k,s,p,cin,cout,H,W", X2_CASES)
def test_conv_split_fp16_matches_oracle(k, s, p, cin, cout, H, W):
    """split-fp16 storage + 3 fp16 MFMAs per product: fp32-grade (22-bit operands, fp32 accumulate)."""
    x = rnd((2, H, W, cin), 0)
    w = rnd((k, k, cin, cout), 1, (2.0 / (k * k * cin)) ** 0.5)
    b = rnd((cout,), 2, 0.1)
    want = refnn.conv2d(x, w, b, stride=s, padding=p, activation=refnn.leaky_relu)
    got = run_conv(x, w, b, "conv", k, s, p, True, "f16x2", cout_off=8, extra_out=8)
    assert run_conv.last_layout == 1
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)


def test_split_fp16_stem_deconv_head():
    # fp32 stem input -> split-fp16 output on the generic kernel (conv1 of a split-fp16 network)
    x = rnd((2, 32, 48, 6), 3)
    w = rnd((7, 7, 6, 64), 4, (2.0 / (49 * 6)) ** 0.5)
    b = rnd((64,), 5, 0.1)
    want = refnn.conv2d(x, w, b, stride=2, padding=3, activation=refnn.leaky_relu)
    got = run_conv(x, w, b, "conv", 7, 2, 3, True, "f16x2", in_f32=True)
    assert run_conv.last_layout == 0
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)
    # transposed conv (4 phases) with split-K
    xd = rnd((1, 3, 4, 512), 12)
    wd = rnd((4, 4, 256, 512), 13, (2.0 / (4 * 512)) ** 0.5)
    wantd = refnn.conv2d_transpose(xd, wd, activation=refnn.leaky_relu)
    gotd = run_conv(xd, wd, None, "deconv", 4, 2, 1, True, "f16x2", cout_off=8)
    np.testing.assert_allclose(gotd, wantd, rtol=2e-5, atol=2e-5)
    # flow head reading split-fp16 activations
    xh = rnd((2, 12, 16, 194), 14)
    wh = rnd((3, 3, 194, 2), 15, 0.05)
    bh = rnd((2,), 16, 0.1)
    wanth = refnn.conv2d(xh, wh, bh, stride=1, padding=1)
    goth = run_conv(xh, wh, bh, "conv", 3, 1, 1, False, "f16x2", out_f32=True)
    np.testing.assert_allclose(goth, wanth, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("dtype", ["f32", "f16x2", "bf16"])
@pytest.mark.parametrize("k,stride,cin,cs", [(7, 2, 3, 8), (7, 2, 12, 16), (3, 1, 6, 8), (3, 1, 11, 16)])
def test_stem_rowrun_conv(dtype, k, stride, cin, cs):
    """kind 2: the network stems (flownet_c.py:30, flownet_s.py:39, flownet_sd.py:29, flownet2.py:61) on their
    pre-padded few-channel input, one contiguous (kw x cs) run per kernel row."""
    from src import _hip, weights as W
    lib = _hip.lib()
    pad = k // 2
    N, H, Wd, cout = 2, 16, 24, 64
    x = rnd((N, H, Wd, cin), 30)
    w = rnd((k, k, cin, cout), 31, (2.0 / (k * k * cin)) ** 0.5)
    b = rnd((cout,), 32, 0.1)
    if dtype == "bf16":
        x = torch.from_numpy(x).bfloat16().float().numpy()
        w = torch.from_numpy(w).bfloat16().float().numpy()
    want = refnn.conv2d(x, w, b, stride=stride, padding=pad, activation=refnn.leaky_relu)
    xp = np.zeros((N, H + 2 * pad, Wd + 2 * pad, cs), np.float32)
    xp[:, pad:pad + H, pad:pad + Wd, :cin] = x
    xin = _to_dev(xp, dtype)
    code = _CODE[dtype]
    esz = 2 if code in (1, 2) else 4
    line = 128 // esz
    run = (k * cs + line - 1) // line * line
    plan = _hip.conv_plan(code, run, cout)
    assert plan.layout == 1
    packed, cin_pad, cout_pad, kpad = W.pack_stem(w, cs, run, plan.cout_tile, plan.layout)
    scale = 1.0
    if plan.wgt_dtype == 3:
        k2 = int(np.floor(np.log2(1024.0 / np.abs(packed).max())))
        packed, scale = packed * 2.0 ** k2, 2.0 ** -k2
    wdev = W.packed_to_device(packed, plan.wgt_dtype, "cuda")
    bdev = torch.from_numpy(b).cuda()
    oh, ow = want.shape[1], want.shape[2]
    out = _to_dev(np.zeros((N, oh, ow, 64), np.float32), dtype)
    d = _hip.Fn2ConvDesc()
    d.inp, d.out = _hip.view(xin, cs, 0, code), _hip.view(out, cout, 0, code)
    d.wgt, d.bias = wdev.data_ptr(), bdev.data_ptr()
    d.kind, d.kh, d.kw, d.stride, d.pad, d.act = 2, k, k, stride, 0, 1
    d.cin_pad, d.cout_pad, d.kpad, d.wgt_layout, d.out_scale = cin_pad, cout_pad, kpad, plan.layout, scale
    _hip.check(lib.fn2_conv2d(C.byref(d), _hip.stream_ptr()))
    torch.cuda.synchronize()
    got = _from_dev(out, code)
    tol = 1e-2 if dtype == "bf16" else 2e-5
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol)


@pytest.mark.parametrize("k,stride,cin,cs,H,Wd,pad", [
    (7, 2, 12, 16, 12, 512, 3),    # FlowNetS conv1 inside the stacks (flownet_s.py:39): 2 tiles per output row
    (7, 2, 6, 8, 10, 256, 3),      # FlowNetS conv1 on a plain pair
    (4, 1, 16, 16, 7, 259, 0),     # FlowNetC conv1 on 2x2 super-pixels (flownet_c.py:30-34)
    (3, 1, 6, 8, 9, 256, 1),       # FlowNetSD conv0 (flownet_sd.py:29)
    (3, 1, 11, 16, 9, 384, 1),     # fuse_conv0 (flownet2.py:61)
])
def test_stem_from_raw_row_segment(k, stride, cin, cs, H, Wd, pad):
    """conv_rowrun_kernel (split fp16, output rows of whole 128-pixel tiles): the MFMA operand gathered from the raw
    row segment in LDS equals the line form (conv_igemm2_kernel) and the oracle."""
    from src import _hip, weights as W
    lib = _hip.lib()
    N, cout = 2, 64
    x = rnd((N, H, Wd, cin), 40)
    w = rnd((k, k, cin, cout), 41, (2.0 / (k * k * cin)) ** 0.5)
    b = rnd((cout,), 42, 0.1)
    want = refnn.conv2d(x, w, b, stride=stride, padding=pad, activation=refnn.leaky_relu)
    xp = np.zeros((N, H + 2 * pad, Wd + 2 * pad, cs), np.float32)
    xp[:, pad:pad + H, pad:pad + Wd, :cin] = x
    if cs > cin:  # finite garbage in the padding channels meets zero weights
        xp[..., cin:] = 3.0
    xin = _to_dev(xp, "f16x2")
    code = _CODE["f16x2"]
    run = (k * cs + 31) // 32 * 32
    plan = _hip.conv_plan(code, run, cout)
    packed, cin_pad, cout_pad, kpad = W.pack_stem(w, cs, run, plan.cout_tile, plan.layout)
    k2 = int(np.floor(np.log2(1024.0 / np.abs(packed).max())))
    packed, scale = packed * 2.0 ** k2, 2.0 ** -k2
    wdev = W.packed_to_device(packed, plan.wgt_dtype, "cuda")
    bdev = torch.from_numpy(b).cuda()
    oh, ow = want.shape[1], want.shape[2]
    assert ow % 128 == 0
    out = _to_dev(np.zeros((N, oh, ow, 64), np.float32), "f16x2")
    d = _hip.Fn2ConvDesc()
    d.inp, d.out = _hip.view(xin, cs, 0, code), _hip.view(out, cout, 0, code)
    d.wgt, d.bias = wdev.data_ptr(), bdev.data_ptr()
    d.kind, d.kh, d.kw, d.stride, d.pad, d.act = 2, k, k, stride, 0, 1
    d.cin_pad, d.cout_pad, d.kpad, d.wgt_layout, d.out_scale = cin_pad, cout_pad, kpad, plan.layout, scale
    name = C.create_string_buffer(256)
    _hip.check(lib.fn2_conv2d_kernel_name(C.byref(d), name, 256))
    assert name.value.decode().startswith("conv_rowrun_kernel"), name.value
    _hip.check(lib.fn2_conv2d(C.byref(d), _hip.stream_ptr()))
    torch.cuda.synchronize()
    got = _from_dev(out, code)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)


def test_upsample_flow_matches_oracle():
    from src import _hip
    lib = _hip.lib()
    x = rnd((2, 5, 7, 2), 7)
    w = rnd((4, 4, 2, 2), 8, 0.5)
    want = refnn.conv2d_transpose(x, w)
    out = torch.full((2, 10, 14, 8), 3.0, device="cuda")
    v = _hip.view(out, 2, 4)
    xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()  # keep alive across the async launch
    _hip.check(lib.fn2_upsample_flow(_hip.ptr(xd), _hip.ptr(wd), None, C.byref(v), 2, 5, 7, _hip.stream_ptr()))
    res = out.cpu().numpy()
    np.testing.assert_allclose(res[..., 4:6], want, rtol=1e-5, atol=1e-5)
    assert np.all(res[..., :4] == 3.0) and np.all(res[..., 6:] == 3.0)
    # with a bias: the FlowNet2 fusion net's fuse_upsample_flow2to1 / 1to0 (flownet2.py:70-73, :86-89)
    bias = np.array([0.375, -1.25], np.float32)
    bd = torch.from_numpy(bias).cuda()
    _hip.check(lib.fn2_upsample_flow(_hip.ptr(xd), _hip.ptr(wd), _hip.ptr(bd), C.byref(v), 2, 5, 7, _hip.stream_ptr()))
    np.testing.assert_allclose(out.cpu().numpy()[..., 4:6], refnn.conv2d_transpose(x, w, bias=bias), rtol=1e-5, atol=1e-5)


def test_conv_rejects_bad_descriptors():
    from src import _hip
    lib = _hip.lib()
    d = _hip.Fn2ConvDesc()
    assert lib.fn2_conv2d(C.byref(d), None) == _hip.ERR_INVALID_ARGUMENT
    with pytest.raises(ValueError):
        _hip.check(lib.fn2_conv2d(C.byref(d), None))


def _random_cases(seed, n):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        k = int(rng.choice([1, 3, 5]))
        s = int(rng.choice([1, 2])) if k > 1 else 1
        cin = int(rng.choice([32, 64, 96, 160, 224, 386]))
        cout = int(rng.choice([8, 18, 32, 48, 64, 100, 128, 192, 320]))
        H, Wd = int(rng.integers(5, 40)), int(rng.integers(5, 70))
        cases.append((k, s, k // 2, cin, cout, H, Wd))
    return cases


@pytest.mark.parametrize("dtype", ["f32", "f16x2"])
def test_conv_random_shapes_cover_every_tile_variant(dtype):
    """Ragged pixel counts (M not a multiple of any tile), channel counts that are not multiples of the cout
    tile, odd stage counts and split-K on every block-tile instantiation of the LDS-DMA kernel."""
    for i, (k, s, p, cin, cout, H, Wd) in enumerate(_random_cases(123, 14)):
        N = 1 + i % 3
        x = rnd((N, H, Wd, cin), 100 + i)
        w = rnd((k, k, cin, cout), 200 + i, (2.0 / (k * k * cin)) ** 0.5)
        b = rnd((cout,), 300 + i, 0.1)
        want = refnn.conv2d(x, w, b, stride=s, padding=p, activation=refnn.leaky_relu)
        got = run_conv(x, w, b, "conv", k, s, p, True, dtype, cout_off=8 if cout % 8 == 0 else 0, extra_out=8)
        np.testing.assert_allclose(got, want, rtol=3e-5, atol=3e-5, err_msg=str((k, s, cin, cout, H, Wd, N)))
    for i, (cin, cout, H, Wd) in enumerate([(96, 32, 7, 9), (224, 64, 5, 33), (386, 128, 9, 6), (64, 200, 12, 20)]):
        x = rnd((2, H, Wd, cin), 400 + i)
        w = rnd((4, 4, cout, cin), 500 + i, (2.0 / (4 * cin)) ** 0.5)
        want = refnn.conv2d_transpose(x, w, activation=refnn.leaky_relu)
        got = run_conv(x, w, None, "deconv", 4, 2, 1, True, dtype, cout_off=8)
        np.testing.assert_allclose(got, want, rtol=3e-5, atol=3e-5, err_msg=str((cin, cout, H, Wd)))


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W", [(3, 384, 512), (90, 50, 70), (2, 6, 8), (1, 96, 128)])
def test_flow_head_gather_small_and_tiled(N, H, W):
    """fn2_flow_head_gather: out[n,y,x,co] = bias[co] + sum_taps t[n,y+ky-1,x+kx-1][(ky*3+kx)*2+co] with zeros outside
    the image (the second half of predict_flowN, flownet_s.py:54-56) -- both launch forms (LDS-tiled from 1024 tiles
    up, thread-per-pixel below), ragged tile edges included; same tap order, so the two agree bit for bit."""
    from src import _hip
    lib = _hip.lib()
    rng = np.random.default_rng(N * 1000 + H)
    t = rng.standard_normal((N, H, W, 32)).astype(np.float32)
    bias = np.array([0.25, -1.5], np.float32)
    want = np.zeros((N, H, W, 2), np.float32) + bias
    tp = np.pad(t, ((0, 0), (1, 1), (1, 1), (0, 0)))
    for ky in range(3):          # fp32 accumulation in the kernel's order
        for kx in range(3):
            j = (ky * 3 + kx) * 2
            want = want + tp[:, ky:ky + H, kx:kx + W, j:j + 2]
    td, bd = torch.from_numpy(t).cuda(), torch.from_numpy(bias).cuda()
    out = torch.full((N, H, W, 2), float("nan"), dtype=torch.float32, device="cuda")
    _hip.check(lib.fn2_flow_head_gather(td.data_ptr(), 32, bd.data_ptr(), out.data_ptr(), N, H, W, _hip.stream_ptr()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("taps", [3, 5])
@pytest.mark.parametrize("N,H,W", [(2, 384, 512), (4, 48, 64), (3, 6, 8), (2, 13, 37)])
@pytest.mark.parametrize("up_dtype", [None, "f16x2", "f32"])
def test_flow_head_tail(taps, N, H, W, up_dtype):
    """fn2_flow_head_tail: pf = bias + the taps x taps shifted partials (zero outside the image), then -- same launch --
    upsample_flowXtoY of it (4x4 stride-2 transposed conv, flownet_s.py:60-63; with a bias as in the fusion net,
    flownet2.py:70-73) into a 2-channel slice.  Both tile forms (8-row tiles from 256 tiles up, 4 x 32 below), ragged
    edges, and `ring`: the outermost pixel ring is taken from pf as it was, not gathered."""
    from src import _hip, weights as Wt
    lib = _hip.lib()
    rng = np.random.default_rng(N * 1000 + H + taps)
    cs = 64 if taps == 5 else 32
    t = rng.standard_normal((N, H, W, cs)).astype(np.float32)
    bias = np.array([0.25, -1.5], np.float32)
    r = taps // 2
    want = np.zeros((N, H, W, 2), np.float32) + bias
    tp = np.pad(t, ((0, 0), (r, r), (r, r), (0, 0)))
    for ky in range(taps):          # fp32 accumulation in the kernel's order
        for kx in range(taps):
            j = (ky * taps + kx) * 2
            want = want + tp[:, ky:ky + H, kx:kx + W, j:j + 2]
    for ring in (0, 1):
        pf0 = rng.standard_normal((N, H, W, 2)).astype(np.float32)
        exp = want.copy()
        if ring:  # the border ring keeps what pf held
            m = np.zeros((H, W), bool)
            m[0], m[-1], m[:, 0], m[:, -1] = True, True, True, True
            exp[:, m] = pf0[:, m]
        td, bd = torch.from_numpy(t).cuda(), torch.from_numpy(bias).cuda()
        pf = torch.from_numpy(pf0).cuda()
        if up_dtype is None:
            _hip.check(lib.fn2_flow_head_tail(td.data_ptr(), cs, taps, bd.data_ptr(), pf.data_ptr(), N, H, W, ring,
                                              None, None, None, _hip.stream_ptr()))
            torch.cuda.synchronize()
            np.testing.assert_array_equal(pf.cpu().numpy(), exp)
            continue
        uw = rnd((4, 4, 2, 2), 8, 0.5)
        ub = np.array([0.375, -1.25], np.float32)
        code = _CODE[up_dtype]
        out = _to_dev(np.full((N, 2 * H, 2 * W, 16), 3.0, np.float32), up_dtype)
        v = _hip.view(out, 2, 10, code)
        uwd, ubd = torch.from_numpy(uw).cuda(), torch.from_numpy(ub).cuda()
        _hip.check(lib.fn2_flow_head_tail(td.data_ptr(), cs, taps, bd.data_ptr(), pf.data_ptr(), N, H, W, ring,
                                          uwd.data_ptr(), ubd.data_ptr(), C.byref(v), _hip.stream_ptr()))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(pf.cpu().numpy(), exp)
        res = _from_dev(out, code)
        np.testing.assert_allclose(res[..., 10:12], refnn.conv2d_transpose(exp, uw, bias=ub), rtol=1e-5, atol=1e-5)
        assert np.all(res[..., :10] == 3.0) and np.all(res[..., 12:] == 3.0)


@pytest.mark.gpu
@pytest.mark.parametrize("taps,N,H,W,nslab", [(3, 2, 24, 40, 3), (5, 1, 13, 70, 4), (3, 1, 64, 512, 2)])
def test_flow_head_tail_sums_raw_split_k_slabs(taps, N, H, W, nslab):
    """fn2_flow_head_tail_slabs: the partial tensor given as the raw split-K slabs of the GEMM that made it
    (fn2_conv_desc.raw_partials) -- scale * (slab 0 + slab 1 + ...), summed in split order -- equals, bit for bit, the tail
    on the tensor a finalize pass would have stored."""
    from src import _hip
    lib = _hip.lib()
    rng = np.random.default_rng(taps * 100 + nslab)
    cs = 2 * taps * taps + 2          # round_up(Cout, 4): 20 / 52
    slabs = rng.standard_normal((nslab, N, H, W, cs)).astype(np.float32)
    scale = np.float32(0.125)
    summed = slabs[0].copy()
    for sl in range(1, nslab):
        summed = summed + slabs[sl]
    summed = summed * scale
    bias = torch.tensor([0.5, -0.25], device="cuda")
    pf_a = torch.zeros((N, H, W, 2), device="cuda")
    pf_b = torch.zeros_like(pf_a)
    sd, td = torch.from_numpy(slabs).cuda(), torch.from_numpy(summed).cuda()
    _hip.check(lib.fn2_flow_head_tail(td.data_ptr(), cs, taps, bias.data_ptr(), pf_a.data_ptr(), N, H, W, 0, None, None, None,
                                      _hip.stream_ptr()))
    _hip.check(lib.fn2_flow_head_tail_slabs(sd.data_ptr(), cs, nslab, N * H * W * cs, C.c_float(float(scale)), taps,
                                            bias.data_ptr(), pf_b.data_ptr(), N, H, W, 0, None, None, None, _hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(pf_a, pf_b)
    with pytest.raises(ValueError):
        _hip.check(lib.fn2_flow_head_tail_slabs(sd.data_ptr(), cs, nslab, 5, C.c_float(1.0), taps, bias.data_ptr(),
                                                pf_b.data_ptr(), N, H, W, 0, None, None, None, _hip.stream_ptr()))


@pytest.mark.gpu
@pytest.mark.parametrize("out_f32", [False, True])
@pytest.mark.parametrize("kind,k,s,p,cin,cout,N,H,W", [
    ("conv", 5, 2, 2, 64, 128, 2, 48, 64),      # conv2: 5x5 stride 2 (flownet_s.py:40)
    ("conv", 3, 1, 1, 256, 256, 2, 24, 32),     # conv3_1
    ("conv", 3, 2, 1, 256, 512, 1, 24, 32),     # conv4: split-K at this size (slabs + finalize)
    ("conv", 3, 1, 1, 473, 200, 1, 13, 19),     # ragged pixels, couts past the view's end, odd stage count, Cin 473
    ("conv", 1, 1, 0, 96, 130, 3, 9, 7),        # 1x1, three stages, second cout tile almost empty
    ("deconv", 4, 2, 1, 386, 128, 2, 12, 16),   # four phases (flownet_s.py:89 deconv2 has 64; 128 takes the 128-cout tile)
    ("deconv", 4, 2, 1, 770, 160, 1, 6, 8),     # phases + split-K
    ("conv", 3, 2, 1, 64, 64, 2, 48, 64),       # 64-cout tile (two waves per 32-cout tile): FlowNetSD conv1 (flownet_sd.py:30)
    ("deconv", 4, 2, 1, 386, 64, 2, 12, 16),    # deconv2 (flownet_s.py:89)
    ("conv", 3, 1, 1, 162, 40, 1, 9, 21),       # 64-cout tile, ragged everything
])
def test_conv_fragment_order_weights(kind, k, s, p, cin, cout, N, H, W, out_f32):
    """wgt_layout 2 (conv2.hip, WREG): the weight operand in MFMA-fragment order, loaded straight into registers by the
    wave that owns the 32 couts; pixel rows through LDS-DMA.  Same arithmetic as layout 1: compared with the oracle and
    with the layout-1 launch (another tile / split choice: fp32 summation order only)."""
    x = rnd((N, H, W, cin), 70)
    if kind == "conv":
        w = rnd((k, k, cin, cout), 71, (2.0 / (k * k * cin)) ** 0.5)
        b = rnd((cout,), 72, 0.1)
        want = refnn.conv2d(x, w, b, stride=s, padding=p, activation=refnn.leaky_relu)
    else:
        w = rnd((4, 4, cout, cin), 71, (2.0 / (4 * cin)) ** 0.5)
        b = None
        want = refnn.conv2d_transpose(x, w, activation=refnn.leaky_relu)
    got = run_conv(x, w, b, kind, k, s, p, True, "f16x2", out_f32=out_f32, cout_off=8, extra_out=8, fragments=True)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)
    base = run_conv(x, w, b, kind, k, s, p, True, "f16x2", out_f32=out_f32, cout_off=8, extra_out=8)
    np.testing.assert_allclose(got, base, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,cin,blocks", [(2, 37, 61, 82, None), (1, 8, 56, 162, None), (3, 5, 9, 32, None),
                                               (1, 70, 130, 82, None), (1, 70, 130, 82, 8), (2, 33, 64, 162, 6),
                                               (1, 3, 3, 82, None)])
def test_flow_head5_fused(N, H, W, cin, blocks, monkeypatch):
    """fn2_flow_head5: a 5x5 two-output convolution (the composed interconvN + predict_flowN head) with the 50 partials
    of a position formed on the matrix cores and summed from LDS -- against the oracle's conv2d with zero padding 2
    (ring = 0), and with ring = 1 the outermost pixel ring left exactly as pf held it.  96- / 192-channel runs take the
    strip form (a block walks down 60 output columns; `blocks` = FN2_H5_BLOCKS, the grid the row segments are cut for:
    one-row segments, several strips, many rows per block), 32 channels the tile form."""
    from src import _hip, weights as Wt
    if blocks is not None:
        monkeypatch.setenv("FN2_H5_BLOCKS", str(blocks))
    lib = _hip.lib()
    x = rnd((N, H, W, cin), 80)
    w5 = rnd((5, 5, cin, 2), 81, (1.0 / (25 * cin)) ** 0.5)
    bias = np.array([0.3, -0.7], np.float32)
    want = refnn.conv2d(x, w5, bias, stride=1, padding=2)
    cs = (cin + 31) // 32 * 32
    xp = np.zeros((N, H, W, cs), np.float32)
    xp[..., :cin] = x
    xin = _to_dev(xp, "f16x2")
    plan = _hip.conv_plan(3, cs, 50)
    assert plan.layout == 1 and plan.cout_tile == 64
    w1x1 = np.ascontiguousarray(w5.transpose(2, 0, 1, 3)).reshape(1, 1, cin, 50)
    packed, cin_pad, cout_pad, kpad = Wt.pack_conv(w1x1, plan.cout_tile, plan.kstep_elems, cs, plan.layout)
    k2 = int(np.floor(np.log2(1024.0 / np.abs(packed).max())))
    wdev = Wt.packed_to_device(packed * 2.0 ** k2, plan.wgt_dtype, "cuda")
    bd = torch.from_numpy(bias).cuda()
    v = _hip.view(xin, cin, 0, 3)
    for ring in (0, 1):
        pf0 = rnd((N, H, W, 2), 82)
        pf = torch.from_numpy(pf0).cuda()
        _hip.check(lib.fn2_flow_head5(C.byref(v), wdev.data_ptr(), cin_pad, kpad, C.c_float(2.0 ** -k2), bd.data_ptr(),
                                      pf.data_ptr(), ring, None, None, _hip.stream_ptr()))
        torch.cuda.synchronize()
        exp = want.copy()
        if ring:
            m = np.zeros((H, W), bool)
            m[0], m[-1], m[:, 0], m[:, -1] = True, True, True, True
            exp[:, m] = pf0[:, m]
        np.testing.assert_allclose(pf.cpu().numpy(), exp, rtol=2e-5, atol=2e-5)
    # the ring as extra blocks of the launch: nine weight sets (here: case c scaled by 1 + c / 10), one bias pair per case
    groups = (cin + 7) // 8
    wc = np.zeros((9, 25, groups * 8, 2), np.float32)
    bc = np.zeros((9, 2), np.float32)
    for c in range(9):
        wc[c, :, :cin] = (w5 * (1.0 + c / 10.0)).reshape(25, cin, 2)
        bc[c] = bias + c
    wcd, bcd = torch.from_numpy(wc).cuda(), torch.from_numpy(bc).cuda()
    pf = torch.zeros((N, H, W, 2), dtype=torch.float32, device="cuda")
    _hip.check(lib.fn2_flow_head5(C.byref(v), wdev.data_ptr(), cin_pad, kpad, C.c_float(2.0 ** -k2), bd.data_ptr(),
                                  pf.data_ptr(), 1, wcd.data_ptr(), bcd.data_ptr(), _hip.stream_ptr()))
    torch.cuda.synchronize()
    exp = want.copy()
    lin = want - bias
    for y in range(H):
        for xx in range(W):
            cy = 0 if y == 0 else (2 if y == H - 1 else 1)
            cx = 0 if xx == 0 else (2 if xx == W - 1 else 1)
            c = 3 * cy + cx
            if c != 4:
                exp[:, y, xx] = lin[:, y, xx] * (1.0 + c / 10.0) + bias + c
    np.testing.assert_allclose(pf.cpu().numpy(), exp, rtol=3e-5, atol=3e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,N,H,W", [(162, 16, 2, 6, 128), (128, 32, 1, 5, 256), (96, 16, 3, 3, 128)])
def test_deconv_merged_column_phases(cin, cout, N, H, W):
    """fn2_conv2d kind 5: the transposed conv k4 s2 crop 1 (with bias + LeakyReLU: the fusion net's fuse_deconv0 / 1,
    flownet2.py:66-84) with both column phases of an output row computed by one block -- against the oracle and the
    four-phase kind 1 launch."""
    from src import _hip, weights as Wt
    lib = _hip.lib()
    x = rnd((N, H, W, cin), 90)
    w = rnd((4, 4, cout, cin), 91, (2.0 / (4 * cin)) ** 0.5)
    b = rnd((cout,), 92, 0.5)
    want = refnn.conv2d_transpose(x, w, activation=refnn.leaky_relu, bias=b)
    cs = (cin + 31) // 32 * 32
    xp = np.zeros((N, H, W, cs), np.float32)
    xp[..., :cin] = x
    xin = _to_dev(xp, "f16x2")
    plan = _hip.conv_plan(3, cs, cout)
    assert plan.layout == 1 and plan.cout_tile == 32
    packed, cin_pad, cout_pad, kpad = Wt.pack_deconv_merged(w, plan.cout_tile, plan.kstep_elems, cs, plan.layout)
    k2 = int(np.floor(np.log2(1024.0 / np.abs(packed).max())))
    wdev = Wt.packed_to_device(packed * 2.0 ** k2, plan.wgt_dtype, "cuda")
    bd = torch.from_numpy(b).cuda()
    out = _to_dev(np.full((N, 2 * H, 2 * W, 96), 7.0, np.float32), "f16x2")
    d = _hip.Fn2ConvDesc()
    d.inp, d.out = _hip.view(xin, cin, 0, 3), _hip.view(out, cout, 64, 3)
    d.wgt, d.bias = wdev.data_ptr(), bd.data_ptr()
    d.kind, d.kh, d.kw, d.stride, d.pad, d.act = 5, 4, 4, 2, 1, 1
    d.cin_pad, d.cout_pad, d.kpad, d.wgt_layout, d.out_scale = cin_pad, cout_pad, kpad, plan.layout, 2.0 ** -k2
    name = C.create_string_buffer(256)
    _hip.check(lib.fn2_conv2d_kernel_name(C.byref(d), name, 256))
    assert name.value.decode().startswith("conv_halo_kernel"), name.value
    _hip.check(lib.fn2_conv2d(C.byref(d), _hip.stream_ptr()))
    torch.cuda.synchronize()
    res = _from_dev(out, 3)
    np.testing.assert_allclose(res[..., 64:64 + cout], want, rtol=2e-5, atol=2e-5)
    assert np.all(res[..., :64] == 7.0) and np.all(res[..., 64 + cout:] == 7.0)
    base = run_conv(x, w, b, "deconv", 4, 2, 1, True, "f16x2")
    np.testing.assert_allclose(res[..., 64:64 + cout], base, rtol=1e-5, atol=1e-5)
