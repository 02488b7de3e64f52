"""Static-plan executor for FlowNetS / C / SD / CS / CSS / 2 on one MI355X.

What the reference builds as a TF1 graph and runs through ``sess.run``
(src/net.py:542-570) is here a flat list of C-ABI calls into libflownet2_hip.so
over preallocated NHWC activation buffers:

  * every conv / transposed conv is one fn2_conv2d launch (implicit GEMM on MFMA)
    with zero-pad, bias, LeakyReLU and the concat fused: a layer writes straight
    into its channel slice of the consumer's concat buffer, so tf.concat, tf.pad,
    antipad and the 4-node LeakyReLU never exist as kernels;
  * correlation (+LeakyReLU, + write into the 473-channel concat), flow_warp +
    brightness error + 12-channel stack, the FlowNet2 fusion input, upsample_flow
    and the final resize are single fused HBM-bound kernels;
  * the plan can be captured once into a hipGraph (``capture()``) and replayed.

torch is only the device-memory container (buffers, streams).  dtype "f32" runs
the fp32 MFMA parity path; "bf16" / "f16" the 16-bit MFMA throughput paths (activations and
weights 16-bit, fp32 accumulate, flow heads / warps / resize in fp32).
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _hip, netdefs, weights as W

# "f16x2": split-fp16 storage (fp16 hi + fp16 lo per value, 3 fp16 MFMAs per product): fp32-grade results
# on the fp16 matrix cores; buffers are float32 containers (4 bytes per channel).
_DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16, "f16x2": torch.float32}
_CODE = {"f32": _hip.FN2_F32, "bf16": _hip.FN2_BF16, "f16": _hip.FN2_F16, "f16x2": _hip.FN2_F16X2}
_TILE_ARGS = {128: "2, 2, 2, 2", 64: "1, 4, 2, 2", 32: "1, 4, 1, 2"}  # conv_igemm2_kernel<.., WC, WP, TCN, TPN> per cout tile


def conv2_kernel_args(tile, m, cout_pad, phases, x2=True):
    """Template arguments <WC, WP, TCN, TPN, STAGES, KG, M16> of the conv_igemm2_kernel instantiation the library picks
    (conv.hip: build_args / wants_bp64; conv2.hip: launch2), as rocprofv3 prints them."""
    dbg = int(os.environ.get("FN2_CONV_DBG", "0"))
    blocks64 = -(-m // 64) * (cout_pad // 128) * phases if tile == 128 else 0
    # M16 (16x16x32 instruction): plain 128 x 64 tiles of large split-fp16 layers without split-K (conv2.hip: launch2)
    big = "FN2_M16_MIN" in os.environ and blocks64 >= max(int(os.environ["FN2_M16_MIN"]), 513)  # off by default
    m16 = "true" if (x2 and not dbg & 512 and (dbg & 1024 or big)) else "false"
    full = dbg & 32
    if tile < 128:
        base = {64: "1, 4, 2, 2, 2, 1", 32: "1, 4, 1, 2, 2, 1"}[tile] if full else {64: "1, 4, 2, 1, 2, 1", 32: "1, 4, 1, 1, 2, 1"}[tile]
        return f"{base}, {m16}"
    blocks = -(-m // 64) * (cout_pad // 128) * phases
    if not full and int(os.environ.get("FN2_KG", "1")):  # K groups (conv.hip: build_args): the 6x8 / 12x16 levels
        if blocks <= int(os.environ.get("FN2_KG3_MAX", "0")):
            return f"2, 2, 2, 1, 2, 3, {m16}"
        if blocks <= int(os.environ.get("FN2_KG2_MAX", "95")):
            return f"2, 2, 2, 1, 2, 2, {m16}"
    if not full and blocks >= int(os.environ.get("FN2_BP64_MIN", "96")):
        # 128 x 64 tiles; the 3-slot ring instantiation (STAGES = 3) for one-round grids without split-K (conv.hip)
        ring = 384 <= blocks <= int(os.environ.get("FN2_RING_MAX", "512"))
        return f"2, 2, 2, 1, 3, 1, {m16}" if ring else f"2, 2, 2, 1, 2, 1, {m16}"
    return f"2, 2, 2, 2, 2, 1, {m16}"


_TNAME = {"f32": "float", "bf16": "__bf16", "f16": "_Float16", "f16x2": "fn2::x2_t"}


def _round_up(x, m):
    return (x + m - 1) // m * m


class BatchTooLarge(ValueError):
    """An activation tensor of the requested batch would reach MAX_TENSOR_BYTES: the LDS-DMA kernels address a tensor
    through one buffer descriptor with 31-bit offsets ("limits").  `fit` = pairs per engine that keep
    THIS tensor under the limit (a later, larger tensor may ask for fewer: callers retry -- Net.model runs the batch in
    chunks of that many pairs)."""

    def __init__(self, name, batch, nbytes, limit):
        self.fit = max(1, int(batch * (limit - 1) // nbytes))
        super().__init__("activation tensor %s of a batch of %d pairs is %d bytes (limit %d): at most %d pairs per engine"
                         % (name, batch, nbytes, limit, self.fit))


class Engine:
    MAX_TENSOR_BYTES = 1 << 31   # fn2_conv2d refuses larger views (conv.hip: FN2_REQUIRE on in_bytes)

    def __init__(self, model, weights, batch, height, width, dtype="f32", device=None, heads_as_gemm=True,
                 no_deconv_biases=None, strict=True, uint8_inputs=False, plain_stems=False, fragment_weights=None):
        """uint8_inputs: the plan starts with two table look-up passes that turn uint8 image bytes (set_inputs_u8) into
        the fp32 [0,1] images -- Net.adapt_x's `/ 255.0` (net.py:338-345) after the host-to-device copy instead of
        before it, byte-identical, a quarter of the bytes over the host link.
        no_deconv_biases: FlowNetS_interp's constructor flag (flownet_s_interp.py:12-14); None = what the weights say
        (no ``FlowNetS/predict_flow6/biases`` entry -> True).  strict: a variable under the model's scopes that no
        layer consumes raises ValueError (optimizer slots and biases the reference graph does not declare excepted:
        its Saver restores graph variables only, so those are ignored -- with a warning -- as the reference ignores them)."""
        if model not in netdefs.MODELS:
            raise ValueError("unknown model %r" % model)
        if dtype not in _DT:
            raise ValueError("dtype must be 'f32', 'bf16', 'f16' or 'f16x2'")
        if height % 64 or width % 64:
            raise ValueError("height and width must be multiples of 64 (pad with Net.adapt_x, net.py:373-388)")
        self.lib = _hip.lib()
        self.device = device if device is not None else _hip.require_device()
        self.model, self.N, self.H, self.W = model, int(batch), int(height), int(width)
        self.dtype_name, self.tdtype = dtype, _DT[dtype]
        self.act_code = _CODE[dtype]
        self.code_of = {}  # data_ptr -> fn2_dtype of a buffer
        self.weights = weights
        if no_deconv_biases is None:
            no_deconv_biases = "FlowNetS/predict_flow6/biases" not in weights
        self.no_deconv_biases = bool(no_deconv_biases)
        self.consumed = set()           # variable names read by some layer
        self.ignored_variables = []     # biases present in `weights` that the reference graph does not declare
        self.ops = []      # (name, fn, args) ; args exclude the trailing stream
        self.kernel_of = []  # per op: the device kernel (template instantiation) that does the work
        # Lanes: launches that do not depend on each other are tagged with different lanes; capture() records every
        # lane on its own stream, so the hipGraph holds them as parallel paths and their launches fill the CUs that the
        # other paths' small grids, tails and launch gaps leave idle.  Lane 0: the main chain; 1: FlowNetSD beside the
        # C -> S -> S chain (flownet2.py:22-23); 2 / 3: the flow-head chain (predict_flowN -> upsample_flow, a few
        # blocks per launch) beside the transposed convs of lane 0 / 1; 4: FlowNetC's second tower.  self.syncs holds
        # (op index, waiter, waited): before op #index is issued, lane `waiter` waits for everything issued on lane
        # `waited` so far.  The list order is a valid sequential order: eager launches ignore lanes and syncs.
        # Scratch (split-K workspace, head GEMM partials) is per lane.  FN2_BRANCHES=0: everything on lane 0.
        self.branch_of = []
        self._branch = 0
        self.desc_branch = []
        self.syncs = []
        self._lanes_on = bool(int(os.environ.get("FN2_BRANCHES", "1")))
        # bit b: lane b enabled (else its launches stay on the parent lane).  Measured (tools/ab_branches.py, same
        # process, interleaved): FlowNet2 b4 5.00 ms without lanes, 4.57 with lane 1, 4.54 with lanes 1 + 4 -- but 4.98
        # with the head lane 2 added (every cross-lane edge of the graph costs a few microseconds of queue hand-off:
        # fine-grained fork/joins eat what two long independent chains gain), and lane 3 (cross-waits between two
        # forked streams, neither of them the capture's origin) crashes hipStreamEndCapture on ROCm 7.2.  FlowNetC b8
        # alone: 1.641 -> 1.597 with lanes 2 + 4.
        # Round 3: without a head lane the heads and upsample_flows of a decoder level ride on the transposed conv's launch
        # (_refine), which beats the lane wherever the heads are the plain / GEMM forms: FlowNetS b8 1.099 -> 1.073 ms, b1
        # 0.405 -> 0.362; FlowNetC b8 1.515 -> 1.500, b1 0.525 -> 0.504 (tower lane kept).  FlowNetSD's composed heads
        # (ring + GEMM + tail per level) still want their lane: b4 1.019 with, 1.055 without.
        default_mask = 0b10011 if model == "FlowNet2" else (0b10101 if model == "FlowNetSD" else 0b10001)
        self._lane_mask = int(os.environ.get("FN2_LANE_MASK", str(default_mask)))
        self.keep = []     # keep ctypes structs / tensors alive
        self.bufs = {}
        self.layer_flops = []  # (name, flop) algorithmic, for roofline accounting
        self.layer_bytes = []  # (name, bytes) algorithmic HBM bytes of the HBM-bound ops
        # (name, bytes) of every convolution: its input slice read once + its output slice written once + its weights
        # read once, in the engine's storage formats (SURVEY.md 8d's accounting per layer) -- what `roofline.traffic`
        # (PMC) is compared with
        self.layer_io_bytes = []
        self.in_a = torch.zeros((self.N, self.H, self.W, 3), dtype=torch.float32, device=self.device)
        self.in_b = torch.zeros_like(self.in_a)
        self.uint8_inputs = bool(uint8_inputs)
        if self.uint8_inputs:
            self.in_a_u8 = torch.zeros((self.N, self.H, self.W, 3), dtype=torch.uint8, device=self.device)
            self.in_b_u8 = torch.zeros_like(self.in_a_u8)
            # lut[i] = float32(float64(i) / 255.0): exactly the host arithmetic of adapt_x; or float32(i) for an image
            # whose max is <= 1 (adapt_x leaves it as it is).  One table per image, rewritten only when its mode changes
            self._lut_div = torch.from_numpy((np.arange(256, dtype=np.float64) / 255.0).astype(np.float32)).to(self.device)
            self._lut_raw = torch.arange(256, dtype=torch.float32, device=self.device)
            self._lut = [self._lut_div.clone(), self._lut_div.clone()]
            self._lut_mode = [True, True]
            cnt = self.in_a_u8.numel()
            for src, lut, dst, nm in ((self.in_a_u8, self._lut[0], self.in_a, "a"), (self.in_b_u8, self._lut[1], self.in_b, "b")):
                self.kernel_of.append("u8_to_f32_lut_kernel")
                self.ops.append((f"input_{nm}/u8_to_f32", self.lib.fn2_u8_to_f32_lut,
                                 (_hip.ptr(src), _hip.ptr(lut), _hip.ptr(dst), cnt)))
                self.branch_of.append(0)
        self.graph = None
        self.conv_descs = []
        self.layers = []   # one record per parameterised layer, in forward order (used by the trainer)
        # flow heads as 1x1 GEMM (18 partial outputs per pixel) + gather; the trainer keeps the dot-product
        # head, whose natural-order weight its backward kernels read
        self.heads_as_gemm = heads_as_gemm
        # plain_stems (the trainer): FlowNetC's conv1 as the plain 7x7 stride-2 row-run convolution on the padded
        # 3-channel images instead of the space-to-depth form, so that its packed weight is the reference variable
        self.plain_stems = bool(plain_stems)
        # fragment_weights: split-fp16 layers with more than 64 output channels whose launch would be the plain two-stage
        # 128 x 64 instantiation keep their weights in MFMA-fragment order (wgt_layout 2) and run the kernel variant that
        # loads them straight into registers (conv2.hip, WREG).  Not for the trainer (the filter-gradient kernels write
        # layout 1).  None: FN2_WREG (default on).
        self.fragment_weights = bool(int(os.environ.get("FN2_WREG", "1"))) if fragment_weights is None else bool(fragment_weights)
        # compose_heads: FlowNetSD's / the fusion net's linear interconvN + predict_flowN pairs run as one composed 5x5
        # head (_composed_head).  Inference engines only (the trainer needs both variables); FN2_COMPOSE=0: off (A/B).
        self.compose_heads = bool(heads_as_gemm) and dtype == "f16x2" and bool(int(os.environ.get("FN2_COMPOSE", "1")))
        self._head_t = None
        self._defer_flow = False    # building a sub-network of a stack whose flow is resized by its consumer (_final_flow)
        self._own_ws = set()        # descriptors that keep a split-K workspace of their own (_head_slabs)
        self._pending_head = None   # a plain flow head waiting for the transposed conv that takes it along (_conv: defer_head)
        self.outputs = self._build()
        assert self._pending_head is None
        self._check_variables(strict)
        self._alloc_workspace()
        self._resolve_kernel_names()

    # ------------------------------------------------------------------ buffers / views
    def _buf(self, name, n, h, w, c, dtype=None, stem=False):
        """Activation buffer.  dtype=torch.float32: dense fp32 (flow heads, final flows).  stem=True: a
        packed network input (few channels, spatially pre-padded by its builder kernel)."""
        # channel stride: whole 128-byte lines (64 channels of a 2-byte format, 32 of a 4-byte one) keep every
        # consumer on the LDS-DMA conv kernel; small stems stay at multiples of 8
        line = 64 if self.act_code in (_hip.FN2_BF16, _hip.FN2_F16) else 32
        cs = (_round_up(c, line) if c > 32 else _round_up(c, 8)) if dtype is None else c
        if dtype is None and not stem and line == 32 and 8 < c <= 32:
            cs = 32  # one line: the 16-channel interconv0 output then feeds predict_flow0 as a head GEMM

        td = self.tdtype if dtype is None else dtype
        nbytes = n * h * w * cs * torch.empty((), dtype=td).element_size()
        if nbytes >= self.MAX_TENSOR_BYTES:
            if self.N <= 1:
                raise ValueError("one %d x %d pair needs a %d-byte tensor (%s): above the %d-byte limit of the kernels"
                                 % (self.H, self.W, nbytes, name, self.MAX_TENSOR_BYTES))
            raise BatchTooLarge(name, self.N, nbytes, self.MAX_TENSOR_BYTES)
        t = torch.zeros((n, h, w, cs), dtype=td, device=self.device)
        assert name not in self.bufs, name
        self.bufs[name] = t
        self.code_of[t.data_ptr()] = self.act_code if dtype is None else _hip.FN2_F32
        return t

    def _code(self, buf):
        # batch slices (buf[:N], buf[N:]) share the storage of a registered buffer
        base = buf.untyped_storage().data_ptr()
        return self.code_of.get(buf.data_ptr(), self.code_of.get(base, _hip.dtype_code(buf)))

    def _v(self, buf, c=None, c0=0):
        return _hip.view(buf, c, c0, self._code(buf))

    def _w(self, key):
        """A variable of the checkpoint, marked as consumed."""
        if key not in self.weights:
            raise KeyError("the weights lack %r, a variable of the reference's %s graph" % (key, self.model))
        self.consumed.add(key)
        return self.weights[key]

    def _bias(self, scope, name, kind, cout):
        """Bias vector of a layer, or None where the reference graph declares none (netdefs.has_bias: the
        biases_initializer=None scopes of flownet_s.py:53 / flownet_c.py:58 / flownet_sd.py:44 -- but NOT the FlowNet2
        fusion net, flownet2.py:50-89 -- and FlowNetS_interp's no_deconv_biases)."""
        key = f"{scope}/{name}/biases"
        if not netdefs.has_bias(self.model, name, kind, self.no_deconv_biases):
            if key in self.weights:
                self.ignored_variables.append(key)
            return None
        b = np.asarray(self._w(key), np.float32).reshape(-1)
        if b.shape[0] != cout:
            raise ValueError("%s has %d entries, the layer has %d outputs" % (key, b.shape[0], cout))
        return W.to_device(b, torch.float32, self.device)

    _SLOT_SUFFIXES = ("/Adam", "/Adam_1", "/Momentum", "/ExponentialMovingAverage")

    def _check_variables(self, strict):
        """Every variable under the model's scopes must have been read by a layer (a silently dropped tensor is how the
        FlowNet2 fusion biases went missing in round 1).  Tolerated: optimizer slots / counters, and biases the
        reference graph does not declare (Caffe's deconvolution biases in a converted .npy: the reference's Saver
        restores graph variables only) -- those are listed in self.ignored_variables and warned about."""
        roots = tuple(sorted({scope.split("/")[0] + "/" for scope, _ in netdefs.model_scopes(self.model)}))
        stray = [k for k in self.weights
                 if k.startswith(roots) and k not in self.consumed and k not in self.ignored_variables
                 and not k.endswith(self._SLOT_SUFFIXES)]
        if self.ignored_variables:
            import warnings
            warnings.warn("%s: %d bias tensors of the checkpoint belong to layers the reference graph builds without "
                          "biases and are ignored, as the reference ignores them (first: %s)"
                          % (self.model, len(self.ignored_variables), self.ignored_variables[0]))
        if stray and strict:
            raise ValueError("%s: %d checkpoint variables are read by no layer (first: %s); pass strict=False to "
                             "ignore them" % (self.model, len(stray), ", ".join(sorted(stray)[:4])))
        self.stray_variables = stray

    def _sync(self, waiter, waited):
        if waiter != waited:
            self.syncs.append((len(self.ops), waiter, waited))

    def _lane(self, lane):
        """Context manager: the launches added inside go to `lane` (lane 0 when lanes are disabled)."""
        eng = self

        class _L:
            def __enter__(self_):
                self_.prev = eng._branch
                eng._branch = lane if (eng._lanes_on and (eng._lane_mask >> lane) & 1) else self_.prev

            def __exit__(self_, *exc):
                eng._branch = self_.prev
        return _L()

    def _head_lane(self):
        hd = self._branch + 2
        return hd if (self._lanes_on and self._branch in (0, 1) and (self._lane_mask >> hd) & 1) else self._branch

    def _op(self, name, fn, *args, kernel=None):
        self.kernel_of.append(kernel if kernel is not None else fn.__name__.replace("fn2_", ""))
        self.ops.append((name, fn, args))
        self.branch_of.append(self._branch)

    # ------------------------------------------------------------------ layers
    def _conv(self, scope, spec, src, dst, up=None, up_after=None, defer_head=False):
        """src/dst: (buffer, c0, c).  One fn2_conv2d launch.  up (flow heads only): (name of the upsample_flowXtoY layer
        that follows the head, its destination slice) -- returns True when that upsample was fused into the head's tail
        launch (the caller then emits no fn2_upsample_flow for it).  up_after (transposed convs only): (layer name,
        predict_flow(N+1) buffer, destination slice) of the upsample_flow(N+1)toN that writes the neighbouring slice of
        the same concat buffer: it rides on this launch (fn2_conv_desc.up_src: the split-K finalize pass does it).
        defer_head (plain flow heads only): build the launch but do not emit it -- the next transposed conv, which reads the
        same tensor, takes it along (fn2_conv_desc.head)."""
        name, kind, k, stride, pad, cin, cout, act = spec
        sbuf, sc0, sc = src
        dbuf, dc0, dc = dst
        assert sc == cin and dc == cout, (scope, name, sc, cin, dc, cout)
        # small heads (6x8 .. 24x32 levels at batch 8): the one-launch wave-per-pixel kernel; the GEMM + gather pair is
        # two launches of a handful of blocks there.  FN2_HEAD_GEMM_MIN = pixels from which the GEMM form is used.
        head_px = dbuf.shape[0] * dbuf.shape[1] * dbuf.shape[2]
        if (self.heads_as_gemm and kind == "conv" and cout == 2 and k == 3 and stride == 1 and pad == 1 and not act
                and dbuf.dtype == torch.float32 and head_px >= int(os.environ.get("FN2_HEAD_GEMM_MIN", "8192"))
                and self._head_gemm(scope, spec, src, dst, up)):
            return up is not None and bool(int(os.environ.get("FN2_FUSE_UPFLOW", "1")))
        wname = f"{scope}/{name}/weights"
        in_code = self._code(sbuf)
        esz = 2 if in_code in (_hip.FN2_BF16, _hip.FN2_F16) else 4
        # channels per tap: whole 128-byte lines when the buffer has room (LDS-DMA kernel), else 8-aligned
        cin_pad = _round_up(cin, 8)
        cin_line = _round_up(cin, 128 // esz)
        if sc0 + cin_line <= sbuf.shape[3] and _hip.conv_plan(in_code, cin_line, cout).layout == 1:
            cin_pad = cin_line
        plan = _hip.conv_plan(in_code, cin_pad, cout)
        tile, layout = plan.cout_tile, plan.layout
        # transposed convs with 16 outputs on wide maps (the fusion net's fuse_deconv0): both column phases of an output row
        # in one block (fn2_conv2d kind 5): 164.6 -> 116.9 us.  With 32 outputs (fuse_deconv1) the four-phase form has no
        # half-empty tile to begin with and the third tap slot only adds work: 39.8 -> 50.6 us, not taken.
        # Inference engines only; FN2_DECONV_MERGED=0: off (A/B)
        merged = (kind != "conv" and self.compose_heads and cout == 16 and in_code == _hip.FN2_F16X2 and layout == 1
                  and tile == 32 and sbuf.shape[2] % 128 == 0 and bool(int(os.environ.get("FN2_DECONV_MERGED", "1"))))
        if kind == "conv":
            packed, cin_pad, cout_pad, kpad = W.pack_conv(self._w(wname), tile, plan.kstep_elems, cin_pad, layout)
        elif merged:
            packed, cin_pad, cout_pad, kpad = W.pack_deconv_merged(self._w(wname), tile, plan.kstep_elems, cin_pad, layout)
        else:
            packed, cin_pad, cout_pad, kpad = W.pack_deconv(self._w(wname), tile, plan.kstep_elems, cin_pad, layout)
        bias = self._bias(scope, name, kind, cout)  # transposed convs: only where the graph declares one (fusion net)
        out_scale = 1.0
        if plan.wgt_dtype == _hip.FN2_F16X2:
            # split fp16: scale the weights by 2^k so that max|w| ~ 1024 (lo parts stay normal fp16 numbers
            # over 4 decades of weight magnitude); the kernel multiplies the accumulator by 2^-k (exact)
            wmax = float(abs(packed).max())
            if wmax > 0:
                k2 = int(math.floor(math.log2(1024.0 / wmax)))
                packed = packed * (2.0 ** k2)
                out_scale = 2.0 ** (-k2)
        wdev = W.packed_to_device(packed, plan.wgt_dtype, self.device)
        d = _hip.Fn2ConvDesc()
        d.inp = self._v(sbuf, sc, sc0)
        d.out = self._v(dbuf, dc, dc0)
        d.wgt = wdev.data_ptr()
        d.bias = bias.data_ptr() if bias is not None else None
        d.kind = 0 if kind == "conv" else (5 if merged else 1)
        d.kh = d.kw = k
        d.stride, d.pad = stride, pad
        d.act = _hip.ACT_LEAKY if act else _hip.ACT_NONE
        d.cin_pad, d.cout_pad, d.kpad = cin_pad, cout_pad, kpad
        d.wgt_layout = layout
        d.out_scale = out_scale
        frag = False
        if (self.fragment_weights and layout == 1 and tile >= int(os.environ.get("FN2_WREG_TILE_MIN", "128"))
                and plan.wgt_dtype == _hip.FN2_F16X2 and self._wants_fragments(d)):
            wdev = W.to_fragment_order(wdev)
            d.wgt, d.wgt_layout, frag = wdev.data_ptr(), 2, True
        self.keep += [d, wdev, bias]
        self.conv_descs.append(d)
        self.desc_branch.append(self._branch)
        up_rec = None
        if self._pending_head is not None and kind != "conv":
            hd = self._pending_head
            assert up_after is not None and hd.inp.data == d.inp.data and hd.inp.c0 == d.inp.c0, (scope, name)
            d.head = C.addressof(hd)
            self._pending_head = None
        if up_after is not None:
            assert kind != "conv" and (not merged or cout == 16)   # (kind 5 with 16 outputs: its epilogue writes the two channels)
            up_name, up_src, (ubuf, uc0, uc) = up_after
            assert ubuf is dbuf and uc == 2
            uw = W.to_device(self._w(f"{scope}/{up_name}/weights"), torch.float32, self.device)
            ub = self._bias(scope, up_name, "deconv", 2)
            d.up_src, d.up_w, d.up_bias, d.up_c0 = up_src.data_ptr(), uw.data_ptr(), (ub.data_ptr() if ub is not None else None), uc0
            uv = self._v(ubuf, uc, uc0)
            self.keep += [uw, ub, uv]
            up_rec = dict(scope=scope, name=up_name, kind="upflow", src=up_src, dst=up_after[2], w=uw, b=ub, view=uv)
        # (layout = how the fp32 packed matrix is ordered -- what the trainer's index maps and filter gradients use;
        # frag = the split-fp16 copy `w` the forward launch reads is that matrix re-tiled into fragment order)
        self.layers.append(dict(scope=scope, name=name, kind=d.kind, k=k, stride=stride, pad=pad, cin=cin, cout=cout,
                                act=bool(act), src=src, dst=dst, desc=d, w=wdev, b=bias, cin_pad=cin_pad,
                                cout_pad=cout_pad, kpad=kpad, layout=layout, tile=tile, kstep=plan.kstep_elems, frag=frag))
        if up_rec is not None:
            self.layers.append(up_rec)
        tn = _TNAME[self.dtype_name] if in_code == self.act_code else "float"
        if kind == "conv" and cout == 2 and k == 3 and stride == 1 and pad == 1:
            kern = f"flow_head_kernel<{tn}>"
        elif layout == 1:
            on = _TNAME[self.dtype_name] if self._code(dbuf) == self.act_code else "float"
            m_px = dbuf.shape[0] * dbuf.shape[1] * dbuf.shape[2] // (4 if kind != "conv" else 1)
            kern = f"conv_igemm2_kernel<{tn}, {on}, {conv2_kernel_args(tile, m_px, cout_pad, 4 if kind != 'conv' else 1, tn == 'fn2::x2_t')}>"
        else:
            shape = {128: "4, 2, 2", 64: "4, 1, 4", 32: "2, 1, 4", 16: "1, 1, 4"}[tile]
            on = _TNAME[self.dtype_name] if self._code(dbuf) == self.act_code else "float"
            kern = f"conv_igemm_kernel<{tn}, {on}, {shape}>"
        if defer_head and kern.startswith("flow_head_kernel"):
            assert self._pending_head is None
            self._pending_head = d
        else:
            self._op(f"{scope}/{name}", self.lib.fn2_conv2d, C.byref(d), kernel=kern)
        n, oh, ow = dbuf.shape[0], dbuf.shape[1], dbuf.shape[2]
        taps = k * k if kind == "conv" else 4
        self.layer_flops.append((f"{scope}/{name}", 2.0 * n * oh * ow * taps * cin * cout))
        osz = 4 if dbuf.dtype == torch.float32 else 2
        self.layer_io_bytes.append((f"{scope}/{name}", float(sbuf.shape[0] * sbuf.shape[1] * sbuf.shape[2] * cin * esz
                                                            + n * oh * ow * cout * osz + k * k * cin * cout * esz)))

    def _head_slabs(self, d, head_t, t_cs):
        """Where the tail of a GEMM-form head finds its partials: (pointer, t_cs, slabs, slab stride, scale).  A head GEMM
        that splits K keeps its raw slabs (fn2_conv_desc.raw_partials, in a workspace of its own) and the tail sums them:
        no finalize launch between the two.  FN2_HEAD_SLABS=0: the finalize pass writes head_t as before (A/B)."""
        need = int(self.lib.fn2_conv2d_workspace_bytes(C.byref(d)))
        if need <= 0 or not int(os.environ.get("FN2_HEAD_SLABS", "1")):
            return _hip.ptr(head_t), t_cs, 1, 0, 1.0
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.device)
        d.workspace, d.workspace_bytes, d.raw_partials = ws.data_ptr(), need, 1
        self.keep.append(ws)
        self._own_ws.add(C.addressof(d))
        splits = int(self.lib.fn2_conv2d_splits(C.byref(d)))
        if splits <= 1:
            d.raw_partials = 0
            return _hip.ptr(head_t), t_cs, 1, 0, 1.0
        ws_cs = (d.out.c + 3) // 4 * 4
        return ws.data_ptr(), ws_cs, splits, d.out.n * d.out.h * d.out.w * ws_cs, float(d.out_scale) or 1.0

    def _wants_fragments(self, d):
        """True when the library would run this layer on the plain two-stage 128 x 64 instantiation (no ring, K groups,
        halo, 128 x 128 tiles): those are the launches the fragment-order variant replaces.  The split-K workspace is
        not allocated yet at build time, so the question is asked with a stand-in workspace of the size the layer wants."""
        need = int(self.lib.fn2_conv2d_workspace_bytes(C.byref(d)))
        d.workspace, d.workspace_bytes = (1, need) if need else (None, 0)   # (never dereferenced: the name query launches nothing)
        buf = C.create_string_buffer(256)
        rc = self.lib.fn2_conv2d_kernel_name(C.byref(d), buf, 256)
        d.workspace, d.workspace_bytes = None, 0
        which = os.environ.get("FN2_WREG_KERNELS", "false>")
        return rc == 0 and buf.value.decode().startswith("conv_igemm2_kernel") and buf.value.decode().endswith(which)

    def _head_gemm(self, scope, spec, src, dst, up=None):
        """predict_flowN as a GEMM: 1x1 convolution with 18 outputs (tap*2 + co) on the LDS-DMA kernel into a shared
        fp32 scratch tensor, then fn2_flow_head_tail: the nine shifted partials summed AND, in the same launch, the
        upsample_flowXtoY that follows the head (up = (layer name, destination slice); flownet_s.py:54-63).  Returns
        False when the input slice has no whole-line run."""
        name, kind, k, stride, pad, cin, cout, act = spec
        sbuf, sc0, sc = src
        pf = dst[0]
        in_code = self._code(sbuf)
        esz = 2 if in_code in (_hip.FN2_BF16, _hip.FN2_F16) else 4
        cin_line = _round_up(cin, 128 // esz)
        if sc0 + cin_line > sbuf.shape[3]:
            return False
        plan = _hip.conv_plan(in_code, cin_line, 18)
        if plan.layout != 1:
            return False
        w = np.asarray(self._w(f"{scope}/{name}/weights"), np.float32)          # [3,3,cin,2] HWIO
        w1 = np.ascontiguousarray(w.transpose(2, 0, 1, 3)).reshape(1, 1, cin, 18)     # [ci][(ky*3+kx)*2+co]
        packed, cin_pad, cout_pad, kpad = W.pack_conv(w1, plan.cout_tile, plan.kstep_elems, cin_line, plan.layout)
        out_scale = 1.0
        if plan.wgt_dtype == _hip.FN2_F16X2:
            wmax = float(abs(packed).max())
            if wmax > 0:
                k2 = int(math.floor(math.log2(1024.0 / wmax)))
                packed, out_scale = packed * (2.0 ** k2), 2.0 ** (-k2)
        wdev = W.packed_to_device(packed, plan.wgt_dtype, self.device)
        bias = self._bias(scope, name, kind, cout)
        if self._head_t is None:
            self._head_t = {}
        if self._branch not in self._head_t:  # one scratch for every head of a branch: its launches are ordered
            self._head_t[self._branch] = torch.zeros((self.N * self.H * self.W, 32), dtype=torch.float32, device=self.device)
        head_t = self._head_t[self._branch]
        n, h, wd = pf.shape[0], pf.shape[1], pf.shape[2]
        d = _hip.Fn2ConvDesc()
        d.inp = self._v(sbuf, sc, sc0)
        d.out = _hip.Fn2Tensor(head_t.data_ptr(), _hip.FN2_F32, n, h, wd, 18, 32, 0)
        d.wgt, d.bias = wdev.data_ptr(), None
        d.kind, d.kh, d.kw, d.stride, d.pad = 0, 1, 1, 1, 0
        d.act = _hip.ACT_NONE
        d.cin_pad, d.cout_pad, d.kpad, d.wgt_layout = cin_pad, cout_pad, kpad, plan.layout
        d.out_scale = out_scale
        self.keep += [d, wdev, bias]
        self.conv_descs.append(d)
        self.desc_branch.append(self._branch)
        tn = _TNAME[self.dtype_name] if in_code == self.act_code else "float"
        self._op(f"{scope}/{name}", self.lib.fn2_conv2d, C.byref(d),
                 kernel=f"conv_igemm2_kernel<{tn}, float, {conv2_kernel_args(plan.cout_tile, 0, cout_pad, 1, tn == 'fn2::x2_t')}>")
        t_ptr, t_cs, nslab, slab, tscale = self._head_slabs(d, head_t, 32)
        if up is not None and int(os.environ.get("FN2_FUSE_UPFLOW", "1")):
            up_name, (ubuf, uc0, uc) = up
            uw = W.to_device(self._w(f"{scope}/{up_name}/weights"), torch.float32, self.device)  # [4,4,2,2] HW-O-I
            ub = self._bias(scope, up_name, "deconv", 2)  # only the fusion net's two (flownet2.py:70-73, :86-89)
            uv = self._v(ubuf, uc, uc0)
            self.keep += [uw, ub, uv]
            self.layers.append(dict(scope=scope, name=up_name, kind="upflow", src=pf, dst=up[1], w=uw, b=ub, view=uv))
            self._op(f"{scope}/{name}/tail+{up_name}", self.lib.fn2_flow_head_tail_slabs, t_ptr, t_cs, nslab, slab,
                     C.c_float(tscale), 3, _hip.ptr(bias) if bias is not None else None, _hip.ptr(pf), n, h, wd, 0, _hip.ptr(uw),
                     _hip.ptr(ub) if ub is not None else None, C.byref(uv), kernel="flow_head_tail")
        else:
            self._op(f"{scope}/{name}/tail", self.lib.fn2_flow_head_tail_slabs, t_ptr, t_cs, nslab, slab, C.c_float(tscale), 3,
                     _hip.ptr(bias) if bias is not None else None, _hip.ptr(pf), n, h, wd, 0, None, None, None,
                     kernel="flow_head_tail")
        self.layer_flops.append((f"{scope}/{name}", 2.0 * n * h * wd * 9 * cin * 2))
        self.layer_io_bytes.append((f"{scope}/{name}", float(n * h * wd * (cin * esz + 2 * 4) + 18 * cin * esz)))
        return True

    def _composed_head(self, scope, ic_spec, pf_spec, src, pf, up=None):
        """interconvN (3x3, linear) + predict_flowN (3x3, linear) of FlowNetSD's decoder and the FlowNet2 fusion net
        (flownet_sd.py:60-64 ..., flownet2.py:74-77, :90-93) as ONE 5x5 two-output convolution of the concat buffer with
        the composed weights (src.weights.compose_interconv_head): the interconv's output -- dec_c channels at the
        level's resolution, read by nothing but the head -- is never formed.  Three launches: the border ring with its
        per-case weights (fn2_flow_head_ring), a 1x1 GEMM to the 50 (tap, output) partials of a pixel, and the tail
        (fn2_flow_head_tail: 25 shifted partials + bias, ring pixels taken from the first launch, + the upsample_flow
        that follows the head).  Returns 0 (nothing emitted) when the input slice does not fit the GEMM kernel, 2 when the
        head took `up` into its tail launch, 1 when the caller still has to emit that upsample_flow."""
        sbuf, sc0, sc = src
        in_code = self._code(sbuf)
        if in_code != _hip.FN2_F16X2 or sc0 % 8 or sbuf.shape[1] < 3 or sbuf.shape[2] < 3:
            return False
        cin_line = _round_up(sc, 32)
        if sc0 + cin_line > sbuf.shape[3]:
            return False
        plan = _hip.conv_plan(in_code, cin_line, 50)
        if plan.layout != 1:
            return False
        ic_name, pf_name = ic_spec[0], pf_spec[0]
        w1, w2 = self._w(f"{scope}/{ic_name}/weights"), self._w(f"{scope}/{pf_name}/weights")
        b1 = self._w(f"{scope}/{ic_name}/biases") if netdefs.has_bias(self.model, ic_name, "conv", self.no_deconv_biases) else None
        b2 = self._w(f"{scope}/{pf_name}/biases") if netdefs.has_bias(self.model, pf_name, "conv", self.no_deconv_biases) else None
        w5, bias5 = W.compose_interconv_head(w1, b1, w2, b2)
        n, h, wd = pf.shape[0], pf.shape[1], pf.shape[2]
        # ---- border ring: fp32 weights [9][25][8 * groups][2], zero on the pad channels
        groups = (sc + 7) // 8
        wc = np.zeros((9, 25, groups * 8, 2), np.float32)
        wc[:, :, :sc, :] = w5.reshape(9, 25, sc, 2)
        wcd = W.to_device(wc, torch.float32, self.device)
        bcd = W.to_device(bias5.astype(np.float32), torch.float32, self.device)
        vin = self._v(sbuf, sc, sc0)
        self.keep += [wcd, bcd, vin]
        # ---- interior: 1x1 GEMM with 50 outputs (tap * 2 + o), then the 25-tap tail
        w1x1 = np.ascontiguousarray(w5[4].transpose(2, 0, 1, 3)).reshape(1, 1, sc, 50).astype(np.float32)
        packed, cin_pad, cout_pad, kpad = W.pack_conv(w1x1, plan.cout_tile, plan.kstep_elems, cin_line, plan.layout)
        out_scale = 1.0
        wmax = float(abs(packed).max())
        if wmax > 0:
            k2 = int(math.floor(math.log2(1024.0 / wmax)))
            packed, out_scale = packed * (2.0 ** k2), 2.0 ** (-k2)
        wdev = W.packed_to_device(packed, plan.wgt_dtype, self.device)
        cm = ic_spec[6]
        if n * h * wd >= int(os.environ.get("FN2_HEAD5_MIN", "65536")) and plan.cout_tile == 64 and kpad == cin_pad:
            # large maps: GEMM and tail in one launch, the partials stay in LDS (fn2_flow_head5); the upsample_flow that
            # follows is then its own launch (the caller's)
            b5 = W.to_device(bias5[4].astype(np.float32), torch.float32, self.device)
            vx = self._v(sbuf, sc, sc0)
            self.keep += [wdev, b5, vx]
            strip = cin_pad in (96, 192) and int(os.environ.get("FN2_H5_STRIP", "1"))   # conv.hip: fn2_flow_head5
            self._op(f"{scope}/{ic_name}+{pf_name}", self.lib.fn2_flow_head5, C.byref(vx), _hip.ptr(wdev), cin_pad, kpad,
                     C.c_float(out_scale), _hip.ptr(b5), _hip.ptr(pf), 1, _hip.ptr(wcd), _hip.ptr(bcd),
                     kernel=("head5_strip_kernel<%d>" % (cin_pad // 96)) if strip else
                     "conv_igemm2_kernel<fn2::x2_t, float, 1, 4, 2, 2, 2, 1, false, false, false, true>")
            self.layer_flops.append((f"{scope}/{ic_name}+{pf_name}", 2.0 * n * h * wd * 9 * (sc * cm + cm * 2)))
            self.layer_io_bytes.append((f"{scope}/{ic_name}+{pf_name}", float(n * h * wd * (sc * 4 + 2 * 4) + 9 * (sc * cm + 2 * cm) * 4)))
            return 1
        self._op(f"{scope}/{pf_name}/ring", self.lib.fn2_flow_head_ring, C.byref(vin), _hip.ptr(wcd), _hip.ptr(bcd), _hip.ptr(pf),
                 kernel="flow_head_ring")
        if self._head_t is None:
            self._head_t = {}
        key = (self._branch, 64)
        if key not in self._head_t:
            self._head_t[key] = torch.zeros((self.N * self.H * self.W, 64), dtype=torch.float32, device=self.device)
        head_t = self._head_t[key]
        d = _hip.Fn2ConvDesc()
        d.inp = self._v(sbuf, sc, sc0)
        d.out = _hip.Fn2Tensor(head_t.data_ptr(), _hip.FN2_F32, n, h, wd, 50, 64, 0)
        d.wgt, d.bias = wdev.data_ptr(), None
        d.kind, d.kh, d.kw, d.stride, d.pad = 0, 1, 1, 1, 0
        d.act = _hip.ACT_NONE
        d.cin_pad, d.cout_pad, d.kpad, d.wgt_layout = cin_pad, cout_pad, kpad, plan.layout
        d.out_scale = out_scale
        b5 = W.to_device(bias5[4].astype(np.float32), torch.float32, self.device)
        self.keep += [d, wdev, b5]
        self.conv_descs.append(d)
        self.desc_branch.append(self._branch)
        self._op(f"{scope}/{ic_name}+{pf_name}", self.lib.fn2_conv2d, C.byref(d), kernel="conv_igemm2_kernel (composed head)")
        t_ptr, t_cs, nslab, slab, tscale = self._head_slabs(d, head_t, 64)
        if up is not None:
            up_name, (ubuf, uc0, uc) = up
            uw = W.to_device(self._w(f"{scope}/{up_name}/weights"), torch.float32, self.device)
            ub = self._bias(scope, up_name, "deconv", 2)
            uv = self._v(ubuf, uc, uc0)
            self.keep += [uw, ub, uv]
            self.layers.append(dict(scope=scope, name=up_name, kind="upflow", src=pf, dst=up[1], w=uw, b=ub, view=uv))
            self._op(f"{scope}/{pf_name}/tail+{up_name}", self.lib.fn2_flow_head_tail_slabs, t_ptr, t_cs, nslab, slab,
                     C.c_float(tscale), 5, _hip.ptr(b5), _hip.ptr(pf), n, h, wd, 1, _hip.ptr(uw),
                     _hip.ptr(ub) if ub is not None else None, C.byref(uv), kernel="flow_head_tail")
        else:
            self._op(f"{scope}/{pf_name}/tail", self.lib.fn2_flow_head_tail_slabs, t_ptr, t_cs, nslab, slab, C.c_float(tscale), 5,
                     _hip.ptr(b5), _hip.ptr(pf), n, h, wd, 1, None, None, None, kernel="flow_head_tail")
        # algorithmic work = what the reference graph does (two 3x3 convolutions), whatever form executes it
        self.layer_flops.append((f"{scope}/{ic_name}+{pf_name}", 2.0 * n * h * wd * 9 * (sc * cm + cm * 2)))
        self.layer_io_bytes.append((f"{scope}/{ic_name}+{pf_name}", float(n * h * wd * (sc * 4 + 2 * 4) + 9 * (sc * cm + 2 * cm) * 4)))
        return 2 if up is not None else 1

    def _conv_stem(self, scope, spec, sbuf, dst, s2d=False):
        """First layer of a network on its pre-padded few-channel input: kind-2 row-run convolution
        (the kw taps x cs channels of a kernel row are one contiguous run of whole 128-byte lines).
        s2d: the input buffer holds 2x2 space-to-depth super-pixels (fn2_pack_image_s2d) and the k x k stride-2
        filter runs as a ceil(k/2)^2 stride-1 filter with re-indexed weights."""
        name, kind, k, stride, pad, cin, cout, act = spec
        dbuf, dc0, dc = dst
        assert kind == "conv" and dc == cout
        k_alg, cin_alg = k, cin
        w_hwio = self._w(f"{scope}/{name}/weights")
        if s2d:
            assert stride == 2 and cin == 3 and sbuf.shape[3] == 16
            k2 = (k + 1) // 2
            w = np.zeros((k2, k2, 16, cout), np.float32)
            for ky in range(k):
                for kx in range(k):
                    w[ky // 2, kx // 2, ((ky & 1) * 2 + (kx & 1)) * 4:((ky & 1) * 2 + (kx & 1)) * 4 + 3] = w_hwio[ky, kx]
            w_hwio, k, stride, cin = w, k2, 1, 16
        cs = sbuf.shape[3]
        in_code = self._code(sbuf)
        esz = 2 if in_code in (_hip.FN2_BF16, _hip.FN2_F16) else 4
        run = _round_up(k * cs, 128 // esz)
        plan = _hip.conv_plan(in_code, run, cout)
        assert plan.layout == 1, (scope, name)
        packed, cin_pad, cout_pad, kpad = W.pack_stem(w_hwio, cs, run, plan.cout_tile, plan.layout)
        bias = self._bias(scope, name, kind, cout)
        out_scale = 1.0
        if plan.wgt_dtype == _hip.FN2_F16X2:
            k2 = int(math.floor(math.log2(1024.0 / float(abs(packed).max()))))
            packed, out_scale = packed * (2.0 ** k2), 2.0 ** (-k2)
        wdev = W.packed_to_device(packed, plan.wgt_dtype, self.device)
        d = _hip.Fn2ConvDesc()
        d.inp = self._v(sbuf, cs, 0)
        d.out = self._v(dbuf, dc, dc0)
        d.wgt = wdev.data_ptr()
        d.bias = bias.data_ptr() if bias is not None else None
        d.kind = 2
        d.kh = d.kw = k
        d.stride, d.pad = stride, 0
        d.act = _hip.ACT_LEAKY if act else _hip.ACT_NONE
        d.cin_pad, d.cout_pad, d.kpad = cin_pad, cout_pad, kpad
        d.wgt_layout = plan.layout
        d.out_scale = out_scale
        self.keep += [d, wdev, bias]
        self.conv_descs.append(d)
        self.desc_branch.append(self._branch)
        self.layers.append(dict(scope=scope, name=name, kind=2, k=k, stride=stride, pad=pad, cin=cin, cout=cout,
                                act=bool(act), src=(sbuf, 0, cs), dst=dst, desc=d, w=wdev, b=bias, cin_pad=cin_pad,
                                cout_pad=cout_pad, kpad=kpad, layout=plan.layout, tile=plan.cout_tile,
                                kstep=plan.kstep_elems, cs=cs))
        tn = _TNAME[self.dtype_name]
        self._op(f"{scope}/{name}", self.lib.fn2_conv2d, C.byref(d),
                 kernel=f"conv_igemm2_kernel<{tn}, {tn}, {conv2_kernel_args(plan.cout_tile, 0, cout_pad, 1, tn == 'fn2::x2_t')}>")
        n, oh, ow = dbuf.shape[0], dbuf.shape[1], dbuf.shape[2]
        self.layer_flops.append((f"{scope}/{name}", 2.0 * n * oh * ow * k_alg * k_alg * cin_alg * cout))
        self.layer_io_bytes.append((f"{scope}/{name}", float(sbuf.numel() * esz + n * oh * ow * cout * esz
                                                            + k_alg * k_alg * cin_alg * cout * esz)))

    def _upflow(self, scope, name, src_f32, dst):
        dbuf, dc0, dc = dst
        w = W.to_device(self._w(f"{scope}/{name}/weights"), torch.float32, self.device)  # [4,4,2,2] HW-O-I
        b = self._bias(scope, name, "deconv", 2)  # only the fusion net's two (flownet2.py:70-73, :86-89)
        v = self._v(dbuf, dc, dc0)
        self.keep += [w, v, b]
        n, h, wd, _ = src_f32.shape
        self.layers.append(dict(scope=scope, name=name, kind="upflow", src=src_f32, dst=dst, w=w, b=b, view=v))
        self._op(f"{scope}/{name}", self.lib.fn2_upsample_flow, _hip.ptr(src_f32), _hip.ptr(w),
                 _hip.ptr(b) if b is not None else None, C.byref(v), n, h, wd)

    def _resize(self, name, src_f32, scale):
        n, h, w, c = src_f32.shape
        dst = self._buf(name, n, self.H, self.W, 2, torch.float32)
        self._op(name, self.lib.fn2_resize_bilinear_f32, _hip.ptr(src_f32), _hip.ptr(dst), n, h, w, c, self.H,
                 self.W, C.c_float(scale))
        return dst

    def _resolve_kernel_names(self):
        """Ask the library which instantiation each fn2_conv2d launch takes (tile, ring, K groups, halo: decided in
        conv.hip / conv2.hip from the geometry, the workspace and the tuning knobs) -- the per-kernel tables of the
        bench and the PMC traffic lookup then carry the names rocprofv3 reports."""
        buf = C.create_string_buffer(256)
        for i, (name, fn, args) in enumerate(self.ops):
            if fn is self.lib.fn2_conv2d:
                if self.lib.fn2_conv2d_kernel_name(args[0], buf, 256) == 0 and buf.value:
                    self.kernel_of[i] = buf.value.decode()

    def _alloc_workspace(self):
        """One fp32 split-K scratch buffer per branch, shared by the branch's layers (its launches are ordered)."""
        branches = self.desc_branch + [0] * (len(self.conv_descs) - len(self.desc_branch))  # the trainer appends descs
        self.workspace = {}
        for br in sorted(set(branches)):
            descs = [d for d, b in zip(self.conv_descs, branches) if b == br and C.addressof(d) not in self._own_ws]
            need = max([int(self.lib.fn2_conv2d_workspace_bytes(C.byref(d))) for d in descs] + [0])
            if need > 0:
                self.workspace[br] = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.device)
                for d in descs:
                    d.workspace = self.workspace[br].data_ptr()
                    d.workspace_bytes = need

    # ------------------------------------------------------------------ sub-networks
    def _refine(self, scope, tag, L, c6_1, cats, interconv):
        """4-level decoder (flownet_s.py:52-104; flownet_sd.py:45-103).  The flow-head chain (predict_flowN ->
        upsample_flow, launches of a few blocks) runs on the head lane beside the transposed conv of the same level."""
        N = c6_1.shape[0]
        preds = {}
        M, Hd = self._branch, self._head_lane()
        h, w = c6_1.shape[1], c6_1.shape[2]
        pf = self._buf(f"{tag}/predict_flow6", N, h, w, 2, torch.float32)
        self._sync(Hd, M)  # conv6_1 is there
        skips, decs = (512, 512, 256, 128), (512, 256, 128, 64)
        # without a head lane a plain flow head rides on the transposed conv that follows it (same input tensor) and its
        # upsample_flow on that launch's finalize pass: the level is ONE fn2_conv2d call.  FN2_HEAD_IN_DECONV=0: off (A/B)
        ride_heads = (Hd == M and self.heads_as_gemm and self.dtype_name == "f16x2"
                      and bool(int(os.environ.get("FN2_UP_IN_DECONV", "1"))) and bool(int(os.environ.get("FN2_HEAD_IN_DECONV", "1"))))
        with self._lane(Hd):
            # (a GEMM-form head takes the upsample_flow that follows it into its tail launch)
            fused = self._conv(scope, L["predict_flow6"], (c6_1, 0, 1024), (pf, 0, 2),
                               up=("upsample_flow6to5", (cats[5], skips[0] + decs[0], 2)), defer_head=ride_heads)
        preds["predict_flow6"] = pf
        cur, cur_c = c6_1, 1024
        for i, (lvl, skip_c, dec_c) in enumerate(zip((5, 4, 3, 2), skips, decs)):
            cat = cats[lvl]
            h, w = cat.shape[1], cat.shape[2]
            # without a head lane the upsample_flow of the level rides on the transposed conv's launch (its split-K
            # finalize pass): one launch less per level on the chain.  FN2_UP_IN_DECONV=0: its own launch (A/B)
            ride = (not fused and Hd == M and self._code(cat) == self.act_code
                    and bool(int(os.environ.get("FN2_UP_IN_DECONV", "1"))))   # (the trainer's forward plan as well)
            self._conv(scope, L[f"deconv{lvl}"], (cur, 0, cur_c), (cat, skip_c, dec_c),
                       up_after=(f"upsample_flow{lvl + 1}to{lvl}", pf, (cat, skip_c + dec_c, 2)) if ride else None)
            if not fused and not ride:
                with self._lane(Hd):
                    self._upflow(scope, f"upsample_flow{lvl + 1}to{lvl}", pf, (cat, skip_c + dec_c, 2))
            self._sync(M, Hd)  # concat complete for the main lane's readers (next deconv / interconv)
            if not interconv:
                self._sync(Hd, M)  # ... and for the head (with an interconv the head waits for that instead)
            cur, cur_c = cat, skip_c + dec_c + 2
            head_src = (cat, 0, cur_c)
            pf = self._buf(f"{tag}/predict_flow{lvl}", N, h, w, 2, torch.float32)
            up = None
            if lvl > 2:
                up = (f"upsample_flow{lvl}to{lvl - 1}", (cats[lvl - 1], skips[i + 1] + decs[i + 1], 2))
            if interconv and self.compose_heads:
                self._sync(Hd, M)
                with self._lane(Hd):
                    composed = self._composed_head(scope, L[f"interconv{lvl}"], L[f"predict_flow{lvl}"], head_src, pf, up=up)
                if composed:
                    fused = composed == 2
                    preds[f"predict_flow{lvl}"] = pf
                    continue
            if interconv:
                ic = self._buf(f"{tag}/interconv{lvl}", N, h, w, dec_c)
                self._conv(scope, L[f"interconv{lvl}"], (cat, 0, cur_c), (ic, 0, dec_c))
                self._sync(Hd, M)
                head_src = (ic, 0, dec_c)
            with self._lane(Hd):
                fused = self._conv(scope, L[f"predict_flow{lvl}"], head_src, (pf, 0, 2), up=up,
                                   defer_head=ride_heads and lvl > 2 and not interconv)
            preds[f"predict_flow{lvl}"] = pf
        return preds

    def _final_flow(self, tag, preds, scale):
        """flow = resize_bilinear(scale * predict_flow2) on the head lane; the main lane then waits for it.  Inside a stack
        (self._defer_flow) the resize is left to the op that consumes the flow -- fn2_stack_input_pf / fn2_fusion_input_pf
        interpolate per pixel and write the flow buffer too -- so the chain is one launch shorter per sub-network."""
        M, Hd = self._branch, self._head_lane()
        if self._defer_flow:
            preds["flow"] = self._buf(f"{tag}/flow", self.N, self.H, self.W, 2, torch.float32)
            preds["_flow_src"] = (preds["predict_flow2"], scale)
            self._sync(M, Hd)   # predict_flow2 came from the head lane: its consumer runs on the main lane
            return preds
        with self._lane(Hd):
            preds["flow"] = self._resize(f"{tag}/flow", preds["predict_flow2"], scale)
        self._sync(M, Hd)
        return preds

    def _alloc_cats(self, tag, N):
        H, W_ = self.H, self.W
        return {5: self._buf(f"{tag}/concat5", N, H // 32, W_ // 32, 1026),
                4: self._buf(f"{tag}/concat4", N, H // 16, W_ // 16, 770),
                3: self._buf(f"{tag}/concat3", N, H // 8, W_ // 8, 386),
                2: self._buf(f"{tag}/concat2", N, H // 4, W_ // 4, 194)}

    def _encoder_tail(self, scope, tag, L, cats):
        """conv4 .. conv6_1 from conv3_1 (already in concat3[0:256]); skips land in the concat buffers."""
        N, H, W_ = self.N, self.H, self.W
        c4 = self._buf(f"{tag}/conv4", N, H // 16, W_ // 16, 512)
        self._conv(scope, L["conv4"], (cats[3], 0, 256), (c4, 0, 512))
        self._conv(scope, L["conv4_1"], (c4, 0, 512), (cats[4], 0, 512))
        c5 = self._buf(f"{tag}/conv5", N, H // 32, W_ // 32, 512)
        self._conv(scope, L["conv5"], (cats[4], 0, 512), (c5, 0, 512))
        self._conv(scope, L["conv5_1"], (c5, 0, 512), (cats[5], 0, 512))
        c6 = self._buf(f"{tag}/conv6", N, H // 64, W_ // 64, 1024)
        self._conv(scope, L["conv6"], (cats[5], 0, 512), (c6, 0, 1024))
        c6_1 = self._buf(f"{tag}/conv6_1", N, H // 64, W_ // 64, 1024)
        self._conv(scope, L["conv6_1"], (c6, 0, 1024), (c6_1, 0, 1024))
        return c6_1

    def _net_s(self, scope, tag, x, cin):
        """FlowNetS.model (flownet_s.py:14-120) on the packed input x (6 or 12 channels)."""
        N, H, W_ = self.N, self.H, self.W
        L = {s[0]: s for s in netdefs.flownet_s_layers(cin)}
        cats = self._alloc_cats(tag, N)
        c1 = self._buf(f"{tag}/conv1", N, H // 2, W_ // 2, 64)
        self._conv_stem(scope, L["conv1"], x, (c1, 0, 64))
        self._conv(scope, L["conv2"], (c1, 0, 64), (cats[2], 0, 128))
        c3 = self._buf(f"{tag}/conv3", N, H // 8, W_ // 8, 256)
        self._conv(scope, L["conv3"], (cats[2], 0, 128), (c3, 0, 256))
        self._conv(scope, L["conv3_1"], (c3, 0, 256), (cats[3], 0, 256))
        c6_1 = self._encoder_tail(scope, tag, L, cats)
        preds = self._refine(scope, tag, L, c6_1, cats, False)
        return self._final_flow(tag, preds, 20.0)  # flownet_s.py:107-111

    def _net_c(self, scope, tag):
        """FlowNetC.model (flownet_c.py:15-125).  conv1 of both towers is one 2N-batch launch."""
        N, H, W_ = self.N, self.H, self.W
        L = {s[0]: s for s in netdefs.flownet_c_layers()}
        cats = self._alloc_cats(tag, N)
        # FN2_C_TOWERS=batch: conv2 / conv3 of BOTH towers as one launch each over 2N images (the level-2 concat buffer gets N
        # more images whose first 128 channels hold tower b's conv2) instead of tower b on a lane of its own
        batch_towers = os.environ.get("FN2_C_TOWERS", "lane") == "batch"
        if batch_towers:
            del self.bufs[f"{tag}/concat2"]
            cat2_all = self._buf(f"{tag}/concat2", 2 * N, H // 4, W_ // 4, 194)
            cats[2] = cat2_all[:N]
        # both towers' images as 2x2 space-to-depth super-pixels with pad(.., 3) baked in (:30-34): the 7x7
        # stride-2 conv1 then runs as a 4x4 stride-1 row-run convolution over 16-channel super-pixels
        c1 = self._buf(f"{tag}/conv1", 2 * N, H // 2, W_ // 2, 64)
        if self.plain_stems:
            x2 = self._buf(f"{tag}/images", 2 * N, H + 6, W_ + 6, 3, stem=True)
            v = self._v(x2, 3, 0)
            self.keep.append(v)
            self._op(f"{tag}/pack_a", self.lib.fn2_pack_image, _hip.ptr(self.in_a), N, C.byref(v), 0, 3)
            self._op(f"{tag}/pack_b", self.lib.fn2_pack_image, _hip.ptr(self.in_b), N, C.byref(v), N, 3)
            self._conv_stem(scope, L["conv1"], x2, (c1, 0, 64))
        else:
            x2 = self._buf(f"{tag}/images", 2 * N, (H + 6) // 2, (W_ + 6) // 2, 16, stem=True)
            v = self._v(x2, 16, 0)
            self.keep.append(v)
            self._op(f"{tag}/pack_a", self.lib.fn2_pack_image_s2d, _hip.ptr(self.in_a), N, H, W_, C.byref(v), 0, 3)
            self._op(f"{tag}/pack_b", self.lib.fn2_pack_image_s2d, _hip.ptr(self.in_b), N, H, W_, C.byref(v), N, 3)
            self._conv_stem(scope, L["conv1"], x2, (c1, 0, 64), s2d=True)
        M, T = self._branch, (4 if (self._lanes_on and self._lane_mask & 16 and not batch_towers) else self._branch)  # second tower (and conv_redir) on lane 4
        if batch_towers:
            self._conv(scope, L["conv2"], (c1, 0, 64), (cat2_all, 0, 128))  # images [0, N): conv_a_2 = the level-2 skip, :105
            c3 = self._buf(f"{tag}/conv_ab_3", 2 * N, H // 8, W_ // 8, 256)
            self._conv(scope, L["conv3"], (cat2_all, 0, 128), (c3, 0, 256))
            c3a, c3b = c3[:N], c3[N:]
        else:
            c2b = self._buf(f"{tag}/conv_b_2", N, H // 4, W_ // 4, 128)
            self._sync(T, M)
            self._conv(scope, L["conv2"], (c1[:N], 0, 64), (cats[2], 0, 128))  # conv_a_2 = the level-2 skip, :105
            with self._lane(T):
                self._conv(scope, L["conv2"], (c1[N:], 0, 64), (c2b, 0, 128))
            c3a = self._buf(f"{tag}/conv_a_3", N, H // 8, W_ // 8, 256)
            c3b = self._buf(f"{tag}/conv_b_3", N, H // 8, W_ // 8, 256)
            self._conv(scope, L["conv3"], (cats[2], 0, 128), (c3a, 0, 256))
            with self._lane(T):
                self._conv(scope, L["conv3"], (c2b, 0, 128), (c3b, 0, 256))
            self._sync(M, T)  # the correlation reads both towers
            self._sync(T, M)  # conv_redir (lane T, beside the correlation) reads conv_a_3
        net = self._buf(f"{tag}/corr_concat", N, H // 8, W_ // 8, 473)  # [conv_redir(32) | corr(441)], :46
        va, vb, vo = self._v(c3a, 256, 0), self._v(c3b, 256, 0), self._v(net, 441, 32)
        self.keep += [va, vb, vo]
        tn = _TNAME[self.dtype_name]
        self._op(f"{tag}/correlation", self.lib.fn2_correlation_fused, C.byref(va), C.byref(vb), C.byref(vo), 20, 2,
                 _hip.ACT_LEAKY, kernel=self._corr_kernel_name(tn, H // 8))  # correlation(a3, b3, 1, 20, 1, 2, 20) + LeakyReLU, :40-41
        # pseudo-layer for the trainer: the cost volume has no parameters but passes gradients to both towers
        self.layers.append(dict(scope=scope, name="correlation", kind="corr", fa=c3a, fb=c3b, dst=(net, 32, 441)))
        self.layer_flops.append((f"{scope}/correlation", 2.0 * N * (H // 8) * (W_ // 8) * 441 * 256))
        # a and b read once, the 441 displacement channels written once, in the reference's fp32 terms
        # (11.71 MB per sample at 48 x 64)
        self.layer_bytes.append((f"{tag}/correlation", 4.0 * N * (H // 8) * (W_ // 8) * (2 * 256 + 441)))
        with self._lane(T):
            self._conv(scope, L["conv_redir"], (c3a, 0, 256), (net, 0, 32))
        self._sync(M, T)
        self._conv(scope, L["conv3_1"], (net, 0, 473), (cats[3], 0, 256))
        c6_1 = self._encoder_tail(scope, tag, L, cats)
        preds = self._refine(scope, tag, L, c6_1, cats, False)
        return self._final_flow(tag, preds, 20.0)  # flownet_c.py:112-116

    def _net_sd(self, scope, tag, x):
        """FlowNetSD.model (flownet_sd.py:14-119)."""
        N, H, W_ = self.N, self.H, self.W
        L = {s[0]: s for s in netdefs.flownet_sd_layers()}
        cats = self._alloc_cats(tag, N)
        c0 = self._buf(f"{tag}/conv0", N, H, W_, 64)
        self._conv_stem(scope, L["conv0"], x, (c0, 0, 64))
        c1 = self._buf(f"{tag}/conv1", N, H // 2, W_ // 2, 64)
        self._conv(scope, L["conv1"], (c0, 0, 64), (c1, 0, 64))
        c1_1 = self._buf(f"{tag}/conv1_1", N, H // 2, W_ // 2, 128)
        self._conv(scope, L["conv1_1"], (c1, 0, 64), (c1_1, 0, 128))
        self._conv(scope, L["conv2"], (c1_1, 0, 128), (cats[2], 0, 128))  # skip = conv2, :97
        c2_1 = self._buf(f"{tag}/conv2_1", N, H // 4, W_ // 4, 128)
        self._conv(scope, L["conv2_1"], (cats[2], 0, 128), (c2_1, 0, 128))
        c3 = self._buf(f"{tag}/conv3", N, H // 8, W_ // 8, 256)
        self._conv(scope, L["conv3"], (c2_1, 0, 128), (c3, 0, 256))
        self._conv(scope, L["conv3_1"], (c3, 0, 256), (cats[3], 0, 256))
        c6_1 = self._encoder_tail(scope, tag, L, cats)
        preds = self._refine(scope, tag, L, c6_1, cats, True)
        return self._final_flow(tag, preds, 0.05)  # flownet_sd.py:106-110

    def _corr_kernel_name(self, tn, h):
        """Device kernel fn2_correlation_fused picks (corr.hip): corr3 (same-parity row pairs) for 16-bit / split-fp16
        features when the feature height is a multiple of 4, else corr2."""
        nl = 4 if self.dtype_name in ("bf16", "f16") else 8
        if self.dtype_name != "f32" and h % 4 == 0 and int(os.environ.get("FN2_CORR3", "1")):
            return f"corr3_kernel<{tn}, {tn}, {nl}>"
        return f"corr2_kernel<{tn}, {tn}, {nl}>"

    def _pair_input(self, tag, pad):
        """[a | b] with the stem's zero border of `pad` pixels baked in (flownet_s.py:24,39)."""
        x = self._buf(f"{tag}/pair", self.N, self.H + 2 * pad, self.W + 2 * pad, 6, stem=True)
        v = self._v(x, 6, 0)
        self.keep.append(v)
        self._op(f"{tag}/pack_pair", self.lib.fn2_pack_pair, _hip.ptr(self.in_a), _hip.ptr(self.in_b), C.byref(v), pad)
        return x

    def _stacked_input(self, tag, preds):
        x = self._buf(f"{tag}/stack", self.N, self.H + 6, self.W + 6, 16, stem=True)
        v = self._v(x, 12, 0)
        self.keep.append(v)
        # flow_warp + brightness error + concat, flownet_cs.py:21-36
        if "_flow_src" in preds:
            pf2, scale = preds["_flow_src"]
            self._op(f"{tag}/stack_input", self.lib.fn2_stack_input_pf, _hip.ptr(self.in_a), _hip.ptr(self.in_b), _hip.ptr(pf2),
                     pf2.shape[1], pf2.shape[2], C.c_float(scale), _hip.ptr(preds["flow"]), C.byref(v), 3, kernel="stack_input")
        else:
            self._op(f"{tag}/stack_input", self.lib.fn2_stack_input, _hip.ptr(self.in_a), _hip.ptr(self.in_b),
                     _hip.ptr(preds["flow"]), C.byref(v), 3)
        return x

    def _sub(self, build, *args):
        """A sub-network whose flow goes to a stack_input / fusion_input of the stack (see _final_flow)."""
        prev, self._defer_flow = self._defer_flow, bool(int(os.environ.get("FN2_FLOW_IN_CONSUMER", "1")))
        try:
            return build(*args)
        finally:
            self._defer_flow = prev

    def _net_cs(self, scope, tag):
        pc = self._sub(self._net_c, scope + "/FlowNetC", tag + "/C")
        return self._net_s(scope + "/FlowNetS", tag + "/S", self._stacked_input(tag + "/S", pc), 12)

    def _net_css(self, scope, tag):
        pcs = self._sub(self._net_cs, scope + "/FlowNetCS", tag + "/CS")
        return self._net_s(scope + "/FlowNetS", tag + "/S", self._stacked_input(tag + "/S", pcs), 12)

    def _net_2(self, scope, tag):
        """FlowNet2.model (flownet2.py:18-105)."""
        N, H, W_ = self.N, self.H, self.W
        fork = len(self.ops)  # FlowNetSD (lane 1) depends on nothing but the images: it forks here, in front of the chain
        css = self._sub(self._net_css, scope + "/FlowNetCSS", tag + "/CSS")
        if self._lanes_on and self._lane_mask & 2:
            # FN2_SD_FORK_OPS = launches of the chain FlowNetSD waits for before it starts (0: it forks in front of the chain)
            fork += min(int(os.environ.get("FN2_SD_FORK_OPS", "0")), len(self.ops) - fork)
            self.syncs.append((fork, 1, 0))
            if self._lane_mask & 8 and int(os.environ.get("FN2_FORK3", "1")):
                # the SD head lane forks from the capture's origin stream as well: a stream that first enters the
                # capture by waiting on ANOTHER forked stream crashed hipStreamEndCapture / hipGraphInstantiate (ROCm 7.2)
                self.syncs.append((fork, 3, 0))
        with self._lane(1):
            sd = self._sub(self._net_sd, scope + "/FlowNetSD", tag + "/SD", self._pair_input(tag + "/SD", 1))
        if self._lanes_on and self._lane_mask & 2:
            self._sync(0, 1)
        L = {s[0]: s for s in netdefs.fusion_layers()}
        xf = self._buf(f"{tag}/fusion_in", N, H + 2, W_ + 2, 16, stem=True)
        v = self._v(xf, 11, 0)
        self.keep.append(v)
        if "_flow_src" in sd and "_flow_src" in css:
            (pf_sd, s_sd), (pf_css, s_css) = sd["_flow_src"], css["_flow_src"]
            self._op(f"{tag}/fusion_input", self.lib.fn2_fusion_input_pf, _hip.ptr(self.in_a), _hip.ptr(self.in_b),
                     _hip.ptr(pf_sd), C.c_float(s_sd), _hip.ptr(sd["flow"]), _hip.ptr(pf_css), C.c_float(s_css),
                     _hip.ptr(css["flow"]), pf_sd.shape[1], pf_sd.shape[2], C.byref(v), 1, kernel="fusion_input")
        else:
            self._op(f"{tag}/fusion_input", self.lib.fn2_fusion_input, _hip.ptr(self.in_a), _hip.ptr(self.in_b),
                     _hip.ptr(sd["flow"]), _hip.ptr(css["flow"]), C.byref(v), 1)
        cat0 = self._buf(f"{tag}/concat0", N, H, W_, 82)
        cat1 = self._buf(f"{tag}/concat1", N, H // 2, W_ // 2, 162)
        self._conv_stem(scope, L["fuse_conv0"], xf, (cat0, 0, 64))
        f1 = self._buf(f"{tag}/fuse_conv1", N, H // 2, W_ // 2, 64)
        self._conv(scope, L["fuse_conv1"], (cat0, 0, 64), (f1, 0, 64))
        self._conv(scope, L["fuse_conv1_1"], (f1, 0, 64), (cat1, 0, 128))
        f2 = self._buf(f"{tag}/fuse_conv2", N, H // 4, W_ // 4, 128)
        self._conv(scope, L["fuse_conv2"], (cat1, 0, 128), (f2, 0, 128))
        f2_1 = self._buf(f"{tag}/fuse_conv2_1", N, H // 4, W_ // 4, 128)
        self._conv(scope, L["fuse_conv2_1"], (f2, 0, 128), (f2_1, 0, 128))
        pf2 = self._buf(f"{tag}/predict_flow2", N, H // 4, W_ // 4, 2, torch.float32)
        M, Hd = self._branch, self._head_lane()
        self._sync(Hd, M)
        with self._lane(Hd):  # head + upsample beside the transposed conv, as in _refine
            fused = self._conv(scope, L["predict_flow2"], (f2_1, 0, 128), (pf2, 0, 2),
                               up=("fuse_upsample_flow2to1", (cat1, 160, 2)))
        self._conv(scope, L["fuse_deconv1"], (f2_1, 0, 128), (cat1, 128, 32))
        if not fused:
            with self._lane(Hd):
                self._upflow(scope, "fuse_upsample_flow2to1", pf2, (cat1, 160, 2))
        self._sync(M, Hd)
        pf1 = self._buf(f"{tag}/predict_flow1", N, H // 2, W_ // 2, 2, torch.float32)
        fused, composed = None, 0
        if self.compose_heads:
            self._sync(Hd, M)
            with self._lane(Hd):
                composed = self._composed_head(scope, L["fuse_interconv1"], L["predict_flow1"], (cat1, 0, 162), pf1,
                                               up=("fuse_upsample_flow1to0", (cat0, 80, 2)))
            fused = composed == 2
        if not composed:
            ic1 = self._buf(f"{tag}/fuse_interconv1", N, H // 2, W_ // 2, 32)
            self._conv(scope, L["fuse_interconv1"], (cat1, 0, 162), (ic1, 0, 32))
            self._sync(Hd, M)
            with self._lane(Hd):
                fused = self._conv(scope, L["predict_flow1"], (ic1, 0, 32), (pf1, 0, 2),
                                   up=("fuse_upsample_flow1to0", (cat0, 80, 2)))
        # (no head lane: fuse_upsample_flow1to0 rides on fuse_deconv0 -- the merged-phase epilogue, or a finalize pass)
        ride0 = (not fused and Hd == M and self.dtype_name == "f16x2" and self.heads_as_gemm
                 and bool(int(os.environ.get("FN2_UP_IN_DECONV", "1"))))
        self._conv(scope, L["fuse_deconv0"], (cat1, 0, 162), (cat0, 64, 16),
                   up_after=("fuse_upsample_flow1to0", pf1, (cat0, 80, 2)) if ride0 else None)
        if not fused and not ride0:
            with self._lane(Hd):
                self._upflow(scope, "fuse_upsample_flow1to0", pf1, (cat0, 80, 2))
        self._sync(M, Hd)
        pf0 = self._buf(f"{tag}/predict_flow0", N, H, W_, 2, torch.float32)
        if not (self.compose_heads and self._composed_head(scope, L["fuse_interconv0"], L["predict_flow0"], (cat0, 0, 82), pf0)):
            ic0 = self._buf(f"{tag}/fuse_interconv0", N, H, W_, 16)
            self._conv(scope, L["fuse_interconv0"], (cat0, 0, 82), (ic0, 0, 16))
            self._conv(scope, L["predict_flow0"], (ic0, 0, 16), (pf0, 0, 2))
        # resize_bilinear to (height, width) of a full-resolution tensor is the identity (flownet2.py:100-101)
        return {"predict_flow0": pf0, "flow": pf0}

    def _build(self):
        m = self.model
        if m in ("FlowNetS", "FlowNetS_interp"):  # interp: second "image" = [0.05*sparse_flow | matches]
            return self._net_s("FlowNetS", "S", self._pair_input("S", 3), 6)
        if m == "FlowNetC":
            return self._net_c("FlowNetC", "C")
        if m == "FlowNetSD":
            return self._net_sd("FlowNetSD", "SD", self._pair_input("SD", 1))
        if m == "FlowNetCS":
            return self._net_cs("FlowNetCS", "CS")
        if m == "FlowNetCSS":
            return self._net_css("FlowNetCSS", "CSS")
        return self._net_2("FlowNet2", "F2")

    # ------------------------------------------------------------------ execution
    def set_inputs(self, input_a, input_b):
        """input_a/b: [N,H,W,3] float32 in [0,1] (torch, any device, or numpy)."""
        for dst, src in ((self.in_a, input_a), (self.in_b, input_b)):
            t = src if isinstance(src, torch.Tensor) else torch.as_tensor(src)
            if tuple(t.shape) != tuple(dst.shape):
                raise ValueError("input shape %s != engine shape %s" % (tuple(t.shape), tuple(dst.shape)))
            dst.copy_(t.to(dtype=torch.float32), non_blocking=True)

    def set_inputs_u8(self, input_a, input_b, scale=(True, True)):
        """uint8 [N,H,W,3] images (host -- ideally pinned -- or device, torch or numpy), already zero-padded to the engine
        size.  scale[i]: divide image i by 255 (adapt_x does when the image's max exceeds 1, net.py:338-345); the
        conversion itself is the first two launches of the plan."""
        if not self.uint8_inputs:
            raise ValueError("engine was built without uint8_inputs=True")
        for i, (dst, src) in enumerate(((self.in_a_u8, input_a), (self.in_b_u8, input_b))):
            t = src if isinstance(src, torch.Tensor) else torch.as_tensor(src)
            if t.dtype != torch.uint8:
                raise ValueError("set_inputs_u8 takes uint8 images, got %s" % t.dtype)
            if tuple(t.shape) != tuple(dst.shape):
                raise ValueError("input shape %s != engine shape %s" % (tuple(t.shape), tuple(dst.shape)))
            dst.copy_(t, non_blocking=True)
            if bool(scale[i]) != self._lut_mode[i]:
                self._lut[i].copy_(self._lut_div if scale[i] else self._lut_raw)
                self._lut_mode[i] = bool(scale[i])

    def set_inputs_interp(self, input_a, matches_a, sparse_flow):
        """FlowNetS_interp input (flownet_s_interp.py:34-38): [image (3) | 0.05 * sparse_flow (2) | matches (1)]."""
        dev = lambda x: (x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))).to(
            device=self.device, dtype=torch.float32)
        sf, m = dev(sparse_flow), dev(matches_a)
        if m.ndim == 3:
            m = m[..., None]
        self.set_inputs(dev(input_a), torch.cat([sf * 0.05, m], dim=3))

    def launch(self):
        """Enqueue one forward pass on torch's current stream (no host sync)."""
        if self.graph is not None:
            _hip.check(self.lib.fn2_graph_launch(self.graph, _hip.stream_ptr()))
            return
        s = _hip.stream_ptr()
        for name, fn, args in self.ops:
            rc = fn(*args, s)
            if rc:
                try:
                    _hip.check(rc)
                except Exception as e:
                    raise type(e)("%s: %s" % (name, e)) from None

    def capture(self):
        """Record the plan into a hipGraph on a side stream; later launch() calls replay it."""
        if self.graph is not None:
            return
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        extra = {b: torch.cuda.Stream(device=self.device) for b in sorted(set(self.branch_of) - {0})}
        with torch.cuda.stream(side):
            self.launch()  # warm: module load, first-touch
            side.synchronize()
            _hip.check(self.lib.fn2_capture_begin(_hip.stream_ptr()))
            try:
                if not extra:
                    self.launch()
                else:
                    self._launch_branches(side, extra)
            finally:
                g = C.c_void_p()
                rc = self.lib.fn2_capture_end(_hip.stream_ptr(), C.byref(g))
            _hip.check(rc)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = g
        self._capture_streams = (side, extra)

    def _launch_branches(self, main, extra):
        """Issue the plan with lane b > 0 on its own stream (inside a stream capture: parallel paths of the graph),
        applying self.syncs; every lane is joined back into `main` at the end.

        The ORDER in which the launches are captured is the order in which a replay submits them (node creation order:
        the runtime walks the graph's nodes on the host, a few microseconds each), so a lane whose launches all sit at the
        end of the list -- FlowNetSD, built after the whole C -> S -> S chain -- would be submitted only after the ~100
        launches in front of it.  FN2_LANE_ORDER: 'list' = the plan's own order; 'fair' (default) = the lanes merged so
        that each has issued the same fraction of its launches; 'lanes' = side lanes first.  A sync (idx, waiter, waited)
        keeps its meaning under any order: the event is recorded on `waited` when that lane has issued all its launches
        in front of list position idx, and `waiter` waits for it before its first launch at or behind idx."""
        streams = dict(extra)
        streams[0] = main
        events = []  # kept alive until the capture has ended (an event destroyed while its record node is being captured
        #              is not something to rely on)
        self._capture_events = events
        mode = os.environ.get("FN2_LANE_ORDER", "fair")
        n = len(self.ops)
        lanes = sorted(set(self.branch_of) | {0})
        per = {l: [i for i in range(n) if self.branch_of[i] == l] for l in lanes}
        ptr = {l: 0 for l in lanes}
        nxt = lambda l: per[l][ptr[l]] if ptr[l] < len(per[l]) else n   # list position of the lane's next launch
        syncs = [dict(idx=idx, waiter=wr, waited=wd, ev=None) for idx, wr, wd in self.syncs]

        def record_ready(lane):
            """Lane `lane` has issued everything in front of nxt(lane): record the events of the syncs that wait for that."""
            for sy in syncs:
                if sy["waited"] == lane and sy["ev"] is None and sy["idx"] <= nxt(lane):
                    ev = torch.cuda.Event()
                    ev.record(streams[lane])
                    events.append(ev)
                    sy["ev"] = ev

        for l in lanes:
            record_ready(l)
        issued = 0
        self._issue_order = []
        while issued < n:
            # lanes whose next launch has all its events recorded
            ready = []
            for l in lanes:
                i = nxt(l)
                if i < n and all(sy["ev"] is not None for sy in syncs if sy["waiter"] == l and sy["idx"] <= i and not sy.get("done")):
                    ready.append(l)
            assert ready, "lane schedule deadlocked (syncs do not form a valid order)"
            if mode == "list":
                l = min(ready, key=nxt)
            elif mode == "lanes":
                l = max(ready)
            else:
                l = min(ready, key=lambda q: (ptr[q] / max(len(per[q]), 1), nxt(q)))
            i = nxt(l)
            for sy in syncs:
                if sy["waiter"] == l and sy["idx"] <= i and not sy.get("done"):
                    streams[l].wait_event(sy["ev"])
                    sy["done"] = True
            name, fn, args = self.ops[i]
            rc = fn(*args, streams[l].cuda_stream)
            if rc:
                try:
                    _hip.check(rc)
                except Exception as e:
                    raise type(e)("%s: %s" % (name, e)) from None
            self._issue_order.append(i)
            ptr[l] += 1
            issued += 1
            record_ready(l)
        for sy in syncs:  # waits behind a lane's last launch (nothing left to order: the final join covers them)
            sy["done"] = True
        for b in extra:
            ev = torch.cuda.Event()
            ev.record(streams[b])
            main.wait_event(ev)
            events.append(ev)

    def __call__(self, input_a, input_b):
        self.set_inputs(input_a, input_b)
        self.launch()
        return self.outputs

    @property
    def flops_per_forward(self):
        return sum(f for _, f in self.layer_flops)

    def __del__(self):
        try:
            if self.graph is not None:
                self.lib.fn2_graph_destroy(self.graph)
        except Exception:
            pass
