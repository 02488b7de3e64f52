#!/usr/bin/env python
"""In-process A/B timing of fn2_conv2d variants (cdna guide rule 24: interleaved rounds, one
process, one device).  Variants are FN2_CONV_DBG bit sets (see conv2.hip): 8 flips the DMA
interleave choice, 16 selects the direct epilogue; 1/2/4 are ablations (wrong results).

  python tools/ab_conv.py --dtype bf16 --variants 0,8,16 --rounds 30
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-tf_amd"))
from src import _hip, weights as W  # noqa: E402

LAYERS = {  # name: (kind, k, stride, pad, cin, cout, N, H, W) -- FlowNetC batch 8 at 512x384
    "conv2": ("conv", 5, 2, 2, 64, 128, 8, 192, 256),
    "conv3": ("conv", 5, 2, 2, 128, 256, 8, 96, 128),
    "conv3_1": ("conv", 3, 1, 1, 256, 256, 8, 48, 64),
    "conv4": ("conv", 3, 2, 1, 256, 512, 8, 48, 64),
    "conv4_1": ("conv", 3, 1, 1, 512, 512, 8, 24, 32),
    "conv5": ("conv", 3, 2, 1, 512, 512, 8, 24, 32),
    "conv5_1": ("conv", 3, 1, 1, 512, 512, 8, 12, 16),
    "conv6": ("conv", 3, 2, 1, 512, 1024, 8, 12, 16),
    "conv6_1": ("conv", 3, 1, 1, 1024, 1024, 8, 6, 8),
    "deconv5": ("deconv", 4, 2, 1, 1024, 512, 8, 6, 8),
    "deconv4": ("deconv", 4, 2, 1, 1026, 256, 8, 12, 16),
    "deconv3": ("deconv", 4, 2, 1, 770, 128, 8, 24, 32),
    "deconv2": ("deconv", 4, 2, 1, 386, 64, 8, 48, 64),
    # FlowNet2 batch 4: the full-resolution fusion layers and the SD stem
    "fuse_interconv0": ("conv", 3, 1, 1, 82, 16, 4, 384, 512),
    "fuse_interconv1": ("conv", 3, 1, 1, 162, 32, 4, 192, 256),
    "fuse_deconv0": ("deconv", 4, 2, 1, 162, 16, 4, 192, 256),
    "fuse_conv1_1": ("conv", 3, 1, 1, 64, 128, 4, 192, 256),
    # the weight-streaming levels at FlowNet2's batch 4
    "conv5_b4": ("conv", 3, 2, 1, 512, 512, 4, 24, 32),
    "conv5_1_b4": ("conv", 3, 1, 1, 512, 512, 4, 12, 16),
    "conv6_b4": ("conv", 3, 2, 1, 512, 1024, 4, 12, 16),
    "conv6_1_b4": ("conv", 3, 1, 1, 1024, 1024, 4, 6, 8),
    "deconv5_b4": ("deconv", 4, 2, 1, 1024, 512, 4, 6, 8),
    "deconv4_b4": ("deconv", 4, 2, 1, 1026, 256, 4, 12, 16),
    "conv4_1_b4": ("conv", 3, 1, 1, 512, 512, 4, 24, 32),
    # long dispatches for counter / clock diagnostics
    "conv3_1_b64": ("conv", 3, 1, 1, 256, 256, 64, 48, 64),
}


def build(name, dtype):
    kind, k, stride, pad, cin, cout, N, H, Wd = LAYERS[name]
    lib = _hip.lib()
    td = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16, "f16x2": torch.float32}[dtype]
    code = {"f32": 0, "bf16": 1, "f16": 2, "f16x2": 3}[dtype]
    rng = np.random.default_rng(0)
    cs_in = (cin + 63) // 64 * 64
    x = torch.zeros((N, H, Wd, cs_in), dtype=td, device="cuda")
    xin = np.zeros((N, H, Wd, cs_in), np.float32)
    xin[..., :cin] = rng.standard_normal((N, H, Wd, cin)).astype(np.float32)
    if dtype == "f16x2":
        x = torch.from_numpy(W.split_f16x2(xin).view(np.float32)).cuda()
    else:
        x = torch.from_numpy(xin).cuda().to(td)
    if kind == "conv":
        w = rng.standard_normal((k, k, cin, cout)).astype(np.float32) * 0.05
        oh, ow = (H + 2 * pad - k) // stride + 1, (Wd + 2 * pad - k) // stride + 1
    else:
        w = rng.standard_normal((4, 4, cout, cin)).astype(np.float32) * 0.05
        oh, ow = 2 * H, 2 * Wd
    esz = 2 if dtype in ("bf16", "f16") else 4
    line = 128 // esz
    cin_line = (cin + line - 1) // line * line
    cin_pad = cin_line if _hip.conv_plan(code, cin_line, cout).layout == 1 else (cin + 7) // 8 * 8
    plan = _hip.conv_plan(code, cin_pad, cout)
    pack = W.pack_conv if kind == "conv" else W.pack_deconv
    packed, cin_pad, cout_pad, kpad = pack(w, plan.cout_tile, plan.kstep_elems, cin_pad, plan.layout)
    layout = plan.layout
    wdev = W.packed_to_device(packed, plan.wgt_dtype, "cuda")
    out = torch.zeros((N, oh, ow, (cout + 63) // 64 * 64), dtype=td, device="cuda")
    d = _hip.Fn2ConvDesc()
    d.inp, d.out = _hip.view(x, cin, 0, code), _hip.view(out, cout, 0, code)
    d.wgt, d.bias = wdev.data_ptr(), None
    d.kind = 0 if kind == "conv" else 1
    d.kh = d.kw = k
    d.stride, d.pad, d.act = stride, pad, 1
    d.cin_pad, d.cout_pad, d.kpad, d.wgt_layout = cin_pad, cout_pad, kpad, layout
    # room for the maximum split-K factor (16) whatever knobs a variant sets later
    need = 16 * N * oh * ow * ((cout + 3) // 4 * 4) * 4
    ws = torch.empty(need // 4, dtype=torch.float32, device="cuda")
    d.workspace, d.workspace_bytes = ws.data_ptr(), need
    taps = k * k if kind == "conv" else 4
    flop = 2.0 * N * oh * ow * taps * cin * cout
    return d, flop, (x, wdev, out, ws)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--variants", default="0,16")
    ap.add_argument("--layers", default=",".join(k for k in LAYERS if not k.endswith(("_b64", "_b4"))))
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--inner", type=int, default=5)
    ap.add_argument("--libs", default="", help="comma list of alternative builds of the library (tools/build_variant.sh); "
                    "a variant 'L<i>:<dbg>' runs build i (0 = the in-tree library)")
    a = ap.parse_args()
    lib = _hip.lib()
    libs = [lib]
    for path in [p for p in a.libs.split(",") if p]:
        l = C.CDLL(os.path.abspath(path))
        for name, (res, args) in _hip.PROTOTYPES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        libs.append(l)
    variants = [v for v in a.variants.split(",")]
    for name in a.layers.split(","):
        for kv in ("FN2_BP64_MIN", "FN2_SPLIT_SLOTS", "FN2_CONV_DBG", "FN2_RING_MAX", "FN2_SPLIT_MINBLOCKS"):
            os.environ.pop(kv, None)
        d, flop, keep = build(name, a.dtype)
        times = {v: [] for v in variants}
        s = _hip.stream_ptr()
        for r in range(a.rounds + 2):
            for v in variants:
                li, dbg = (v[1:].split(":") + ["0"])[:2] if v.startswith("L") else ("0", v)
                # "dbg/KEY=VAL/KEY=VAL": extra environment knobs of the library for this variant
                parts = dbg.split("/")
                dbg = parts[0]
                for kv in ("FN2_BP64_MIN", "FN2_SPLIT_SLOTS", "FN2_RING_MAX", "FN2_SPLIT_MINBLOCKS"):
                    os.environ.pop(kv, None)
                for kv in parts[1:]:
                    k_, v_ = kv.split("=")
                    os.environ[k_] = v_
                os.environ["FN2_CONV_DBG"] = dbg
                run = libs[int(li)].fn2_conv2d
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.inner):
                    _hip.check(run(C.byref(d), s))
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    times[v].append(e0.elapsed_time(e1) / a.inner)
        line = "%-8s %-5s" % (name, a.dtype)
        for v in variants:
            t = np.array(times[v])
            line += " | dbg=%-3s med %.4f min %.4f ms %7.1f TF" % (v, np.median(t), t.min(), flop / np.median(t) / 1e9)
        print(line, flush=True)
    os.environ.pop("FN2_CONV_DBG", None)


if __name__ == "__main__":
    main()
