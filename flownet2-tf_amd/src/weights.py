"""Weights: seeded synthetic initialisation, (de)serialisation under the
reference's variable names, and packing into the layout fn2_conv2d reads.

No trained checkpoints exist offline, so parity and timing
use seeded variance-scaling weights; real weights converted by the reference's
scripts/caffe/convert_caffe_weights_to_npy.py (HWIO / HW-O-I, names
``<scope>/<layer>/weights|biases``) load through ``load_npz`` unchanged.
"""
import os

import numpy as np
import torch

from . import netdefs


def init_weights(model, seed=1234, flow_gain=0.25, bias_std=0.02, head_biases=None):
    """Variance-scaling normal weights (the reference trains from
    slim.variance_scaling_initializer(), flownet_s.py:31): std = sqrt(2/fan_in) for
    LeakyReLU layers so activations stay O(1) through ~25 layers; linear layers
    sqrt(1/fan_in); flow heads scaled by ``flow_gain`` so that 20*predict_flow2
    spans several pixels (flow_warp then sees in- and out-of-range targets).
    Small random biases exercise the bias path (the reference initialises them to 0).
    head_biases: FlowNetS_interp only -- False (its default) = the class default no_deconv_biases=True: neither
    predict_flowN nor deconvN get biases (flownet_s_interp.py:12-14, :78-126); True = no_deconv_biases=False.
    Exactly the variables netdefs.has_bias declares are emitted -- including the biases of the FlowNet2 fusion
    net's four transposed convs (flownet2.py:50-89), non-zero so that a path dropping them fails its test."""
    if head_biases is None:
        head_biases = model != "FlowNetS_interp"
    rng = np.random.default_rng(seed)
    w = {}
    for scope, layers in netdefs.model_scopes(model):
        for (name, kind, k, stride, pad, cin, cout, act) in layers:
            if kind == "conv":
                fan_in = k * k * cin
                std = np.sqrt((2.0 if act else 1.0) / fan_in)
                if name.startswith("predict_flow"):
                    std *= flow_gain
                w[f"{scope}/{name}/weights"] = (rng.standard_normal((k, k, cin, cout)) * std).astype(np.float32)
                b = (rng.standard_normal((cout,)) * bias_std).astype(np.float32)  # drawn even when unused: keeps the
                if netdefs.has_bias(model, name, kind, not head_biases):           # stream of the later layers fixed
                    w[f"{scope}/{name}/biases"] = b
            else:
                fan_in = 4 * cin  # each output pixel of a k4 s2 transposed conv sees 2x2 taps
                std = np.sqrt((2.0 if act else 1.0) / fan_in)
                w[f"{scope}/{name}/weights"] = (rng.standard_normal((k, k, cout, cin)) * std).astype(np.float32)
                if netdefs.has_bias(model, name, kind, not head_biases):
                    # own generator: the weights of every layer stay what earlier rounds' fixtures were made with
                    brng = np.random.default_rng([seed, len(w)])
                    w[f"{scope}/{name}/biases"] = (brng.standard_normal((cout,)) * 5 * bias_std).astype(np.float32)
    return w


def save_npz(path, weights):
    np.savez(path, **{k.replace("/", "__"): v for k, v in weights.items()})


def load_npz(path):
    with np.load(path) as z:
        return {k.replace("__", "/"): z[k] for k in z.files}


def load_npy(path):
    """The `.npy` the reference's Caffe converter writes (scripts/caffe/convert_caffe_weights_to_npy.py:489-496):
    np.save of a dict {<tf variable>/weights (HWIO; transposed convs HW-O-I), <tf variable>/biases}.  Biases of
    transposed convs the reference graph builds with biases_initializer=None (flownet_s.py:53, flownet_c.py:58,
    flownet_sd.py:44) are not graph variables: the engine lists them in Engine.ignored_variables and warns; those of
    the FlowNet2 fusion net (flownet2.py:50-89) ARE variables and are applied."""
    obj = np.load(path, allow_pickle=True)
    if obj.dtype != object or obj.shape != ():
        raise ValueError("%s does not hold a pickled {name: array} dict" % path)
    return {str(k): np.asarray(v, np.float32) for k, v in obj.item().items()}


def load_weights(path):
    """.npz (this build's checkpoints), .npy (the reference converter's output) or a TensorFlow V2 checkpoint
    prefix (``flownet-S.ckpt-0`` with its ``.index`` / ``.data-*`` files: what the reference's Saver restores,
    src/net.py:566-569)."""
    from . import tf_checkpoint
    path = str(path)
    for suffix in (".index", ".data-00000-of-00001", ".meta"):  # any file of the bundle names the bundle
        if path.endswith(suffix) and tf_checkpoint.is_tf_checkpoint(path[:-len(suffix)]):
            path = path[:-len(suffix)]
    if tf_checkpoint.is_tf_checkpoint(path):
        return tf_checkpoint.load_tf_checkpoint(path)
    return load_npy(path) if path.endswith(".npy") else load_npz(path)


def checkpoint_exists(path):
    from . import tf_checkpoint
    path = str(path)
    return tf_checkpoint.is_tf_checkpoint(path) or (os.path.exists(path) and path.endswith(
        (".npz", ".npy", ".index", ".data-00000-of-00001")))


def _round_up(x, m):
    return (x + m - 1) // m * m


def _permute_rows64(rows):
    """Row order of the LDS-DMA kernel (fn2_conv2d_weight_layout == 1): inside every group of 32
    rows, packed row (r & 3) + 8 * (r >> 2) + 4 * h holds output channel 16 * h + r (h in 0..1,
    r in 0..15) -- the 32x32 MFMA accumulator layout read backwards, so that the 16 registers of a
    lane are 16 consecutive output channels."""
    n = rows.shape[0]
    assert n % 32 == 0
    src = np.empty(32, np.int64)
    for h in range(2):
        for r in range(16):
            src[(r & 3) + 8 * (r >> 2) + 4 * h] = 16 * h + r
    full = (np.arange(n // 32)[:, None] * 32 + src[None, :]).reshape(-1)
    return rows[full]


def pack_conv(w_hwio, cout_tile, kstep_elems, cin_pad=None, layout=0, dtype=np.float32):
    """[kh,kw,Cin,Cout] -> [cout_pad][kpad] with k = (tap, channel padded to cin_pad)."""
    kh, kw, cin, cout = w_hwio.shape
    cin_pad = _round_up(cin, 8) if cin_pad is None else cin_pad
    assert cin_pad >= cin and cin_pad % 8 == 0
    cout_pad = _round_up(cout, cout_tile)
    kpad = _round_up(kh * kw * cin_pad, kstep_elems)
    p = np.zeros((cout_pad, kh * kw, cin_pad), dtype)
    p[:cout, :, :cin] = np.transpose(w_hwio, (3, 0, 1, 2)).reshape(cout, kh * kw, cin)
    out = np.zeros((cout_pad, kpad), dtype)
    out[:, :kh * kw * cin_pad] = p.reshape(cout_pad, -1)
    if layout == 1:
        out = _permute_rows64(out)
    return out, cin_pad, cout_pad, kpad


def pack_stem(w_hwio, cs, run_pad, cout_tile, layout=1, dtype=np.float32):
    """kind-2 (row-run) stem: [kh,kw,Cin,Cout] -> [cout_pad][kh*run_pad] with k = ky*run_pad + kx*cs + c
    (cs = channel stride of the pre-padded input buffer, run_pad >= kw*cs whole 128-byte lines)."""
    kh, kw, cin, cout = w_hwio.shape
    assert cin <= cs and run_pad >= kw * cs
    cout_pad = _round_up(cout, cout_tile)
    p = np.zeros((cout_pad, kh, run_pad), dtype)
    wt = np.transpose(w_hwio, (3, 0, 1, 2))  # [cout, kh, kw, cin]
    for kx in range(kw):
        p[:cout, :, kx * cs:kx * cs + cin] = wt[:, :, kx, :]
    out = p.reshape(cout_pad, kh * run_pad)
    if layout == 1:
        out = _permute_rows64(out)
    return out, run_pad, cout_pad, kh * run_pad


def pack_deconv(w_hwoi, cout_tile, kstep_elems, cin_pad=None, layout=0, dtype=np.float32):
    """[4,4,Cout,Cin] (HW-O-I) -> [4 phases][cout_pad][kpad]: phase (a,b) is the 2x2 stride-1
    convolution with taps (ty,tx) <- (ky,kx) = (3-a-2ty, 3-b-2tx) producing output pixels (2y+a, 2x+b)."""
    kh, kw, cout, cin = w_hwoi.shape
    assert kh == 4 and kw == 4
    cin_pad = _round_up(cin, 8) if cin_pad is None else cin_pad
    assert cin_pad >= cin and cin_pad % 8 == 0
    cout_pad = _round_up(cout, cout_tile)
    kpad = _round_up(4 * cin_pad, kstep_elems)
    out = np.zeros((4, cout_pad, kpad), dtype)
    for a in range(2):
        for b in range(2):
            p = np.zeros((cout_pad, 4, cin_pad), dtype)
            for ty in range(2):
                for tx in range(2):
                    p[:cout, ty * 2 + tx, :cin] = w_hwoi[3 - a - 2 * ty, 3 - b - 2 * tx]
            ph = np.zeros((cout_pad, kpad), dtype)
            ph[:, :4 * cin_pad] = p.reshape(cout_pad, -1)
            out[a * 2 + b] = _permute_rows64(ph) if layout == 1 else ph
    return out, cin_pad, cout_pad, kpad


def pack_deconv_merged(w_hwoi, cout_tile, kstep_elems, cin_pad, layout=1, dtype=np.float32):
    """[4,4,Cout,Cin] (HW-O-I), Cout in (16, 32) -> [2 row phases][2 Cout][6 cin_pad]: fn2_conv2d kind 5.  Row phase a holds
    both column phases b of the output row 2y + a as ONE 2 x 3-tap stride-1 convolution with 2 Cout output rows
    (row b Cout + co -> output pixel (2y + a, 2x + b), channel co): tap (ty, kx3) reads input (y - 1 + a + ty, x - 1 + kx3);
    column phase b uses kx3 = b + tx, tx in {0, 1}, with the weight of (ky, kx) = (3 - a - 2 ty, 3 - b - 2 tx) and zeros in
    its third slot.  Half the passes over the input of the four-phase form, and for Cout = 16 no half-empty 32-row tile."""
    kh, kw, cout, cin = w_hwoi.shape
    assert kh == 4 and kw == 4 and cout in (16, 32) and cin_pad >= cin and cin_pad % 8 == 0 and cout_tile == 32
    cout_pad = 2 * cout
    kpad = _round_up(6 * cin_pad, kstep_elems)
    out = np.zeros((2, cout_pad, kpad), dtype)
    for a in range(2):
        p = np.zeros((cout_pad, 6, cin_pad), dtype)
        for b in range(2):
            for ty in range(2):
                for tx in range(2):
                    p[b * cout:(b + 1) * cout, ty * 3 + b + tx, :cin] = w_hwoi[3 - a - 2 * ty, 3 - b - 2 * tx]
        ph = np.zeros((cout_pad, kpad), dtype)
        ph[:, :6 * cin_pad] = p.reshape(cout_pad, -1)
        out[a] = _permute_rows64(ph) if layout == 1 else ph
    return out, cin_pad, cout_pad, kpad


def pack_conv_transpose_s2(w_hwio, p, cout_tile, kstep_elems, cin_pad, layout, dtype=np.float32):
    """Weights of fn2_conv2d kind 3 -- the input gradient of a stride-2 convolution with kernel k, pad p and
    forward weight w_hwio [k,k,Ci,Co]: a transposed convolution whose "input channels" are Co and "output
    channels" Ci.  Phase (a,b), tap (ty,tx) holds W[ky][kx]^T with ky = a+p-2*lo_a-2*ty, lo_a = ceil((a+p-k+1)/2)
    (zero where ky is outside [0,k)).  Returns [4][cout_pad][T*T*cin_pad] with cin_pad >= Co, cout_pad >= Ci."""
    k, _, ci, co = w_hwio.shape
    T = (k + 1) // 2
    cout_pad = _round_up(ci, cout_tile)
    kpad = _round_up(T * T * cin_pad, kstep_elems)
    assert cin_pad >= co

    def ceil_half(v):
        return -((-v) // 2)

    out = np.zeros((4, cout_pad, kpad), dtype)
    for a in range(2):
        lo_a = ceil_half(a + p - k + 1)
        for b in range(2):
            lo_b = ceil_half(b + p - k + 1)
            ph = np.zeros((cout_pad, T * T, cin_pad), dtype)
            for ty in range(T):
                ky = a + p - 2 * lo_a - 2 * ty
                if not 0 <= ky < k:
                    continue
                for tx in range(T):
                    kx = b + p - 2 * lo_b - 2 * tx
                    if 0 <= kx < k:
                        ph[:ci, ty * T + tx, :co] = w_hwio[ky, kx]  # [ci, co]
            full = np.zeros((cout_pad, kpad), dtype)
            full[:, :T * T * cin_pad] = ph.reshape(cout_pad, -1)
            out[a * 2 + b] = _permute_rows64(full) if layout == 1 else full
    return out, cin_pad, cout_pad, kpad


def to_device(arr, dtype, device):
    return torch.from_numpy(np.ascontiguousarray(arr)).to(device=device, dtype=dtype).contiguous()


def split_f16x2(packed):
    """fp32 [..., k] (k % 8 == 0) -> split-fp16 storage: per group of 8 values, 8 fp16 hi parts then 8
    fp16 lo parts (x = hi + lo); returned as float16 [..., 2k] (4 bytes per logical element)."""
    p = np.ascontiguousarray(packed, np.float32)
    assert p.shape[-1] % 8 == 0
    g = p.reshape(p.shape[:-1] + (p.shape[-1] // 8, 8))
    hi = g.astype(np.float16)
    lo = (g - hi.astype(np.float32)).astype(np.float16)
    return np.stack([hi, lo], axis=-2).reshape(p.shape[:-1] + (2 * p.shape[-1],))


def join_f16x2(split):
    """Inverse of split_f16x2 (float16 [..., 2k] -> float32 [..., k])."""
    s = np.asarray(split, np.float16)
    g = s.reshape(s.shape[:-1] + (s.shape[-1] // 16, 2, 8)).astype(np.float32)
    return (g[..., 0, :] + g[..., 1, :]).reshape(s.shape[:-1] + (s.shape[-1] // 2,))


_TORCH_OF_CODE = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}


def compose_interconv_head(w1, b1, w2, b2):
    """A linear 3x3 interconvN (w1 [3,3,Cin,Cm] HWIO, bias b1 [Cm]) followed by the linear 3x3 predict_flowN
    (w2 [3,3,Cm,2], bias b2 [2]; flownet_sd.py:60-64, flownet2.py:74-77, :90-93: activation_fn=None on both, pad(.., 1)
    in front of each) as ONE 5x5 convolution of the interconv's input:
        pf(p) = b2 + sum_t w2[t] . ( b1 + sum_s w1[s] . x(p + t + s - 2) )          (x zero outside the image)
    The reference zero-pads the interconv OUTPUT before the second convolution, so a tap t whose position p + t - 1
    falls outside the image contributes nothing -- not the interconv evaluated there.  That only concerns the image's
    outermost pixel ring; which taps drop out depends on the borders p touches: case = 3 * cy + cx with c = 0 (low
    border), 1 (interior), 2 (high border) per axis.
    Returns (w5 [9 cases][5,5,Cin,2], bias5 [9][2]) in float64; case 4 is the interior."""
    w1, w2 = np.asarray(w1, np.float64), np.asarray(w2, np.float64)
    b1 = np.zeros(w1.shape[3]) if b1 is None else np.asarray(b1, np.float64)
    b2 = np.zeros(2) if b2 is None else np.asarray(b2, np.float64)
    cin = w1.shape[2]
    w5 = np.zeros((9, 5, 5, cin, 2))
    bias5 = np.zeros((9, 2))
    ok = {0: (1, 2), 1: (0, 1, 2), 2: (0, 1)}   # taps of the second conv that stay inside the image, per border case
    for cy in range(3):
        for cx in range(3):
            c = 3 * cy + cx
            bias5[c] = b2
            for ty in ok[cy]:
                for tx in ok[cx]:
                    bias5[c] += b1 @ w2[ty, tx]
                    for sy in range(3):
                        for sx in range(3):
                            w5[c, ty + sy, tx + sx] += w1[sy, sx] @ w2[ty, tx]
    return w5, bias5


def to_fragment_order(wdev):
    """Split-fp16 packed weight on the device, layout 1 ([..., cout_pad, k] rows of 128-byte stages
    [hi g0 | lo g0 | ... | hi g3 | lo g3]) -> wgt_layout 2 of fn2_conv2d: per 32-row tile and stage four 1 KiB blocks
    f = 2 q + part (q = which pair of 8-channel groups, part = hi / lo), each lane-linear: lane 32 h + r holds the
    16-byte chunk 4 q + 2 h + part of row r -- exactly the MFMA A-operand fragment the kernel's wave loads with one
    16-byte-per-lane instruction.  A pure permutation of 16-byte chunks (same size, same values)."""
    assert wdev.dtype == torch.float16 and wdev.shape[-2] % 32 == 0 and (wdev.shape[-1] * 2) % 128 == 0
    lead = tuple(wdev.shape[:-2])
    ct, st = wdev.shape[-2] // 32, wdev.shape[-1] * 2 // 128
    v = wdev.contiguous().view(torch.uint8).reshape(lead + (ct, 32, st, 2, 2, 2, 16))   # [.., ct, r, st, q, h, part, 16 B]
    n = len(lead)
    perm = tuple(range(n)) + (n, n + 2, n + 3, n + 5, n + 4, n + 1, n + 6)            # [.., ct, st, q, part, h, r, 16 B]
    return v.permute(perm).contiguous().view(torch.float16).reshape(wdev.shape)


def packed_to_device(packed, wgt_code, device):
    """Packed fp32 weight -> device tensor of the element type the kernel plan asks for."""
    if wgt_code == 3:
        return torch.from_numpy(split_f16x2(packed)).to(device).contiguous()
    return to_device(packed, _TORCH_OF_CODE[wgt_code], device)
