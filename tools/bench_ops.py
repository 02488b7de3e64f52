#!/usr/bin/env python
"""Achieved HBM bandwidth of the HBM-bound custom ops at the BASELINE sizes (algorithmic
bytes per sample), events on the launch stream, median of `--rounds` rounds of `--inner` launches.

  python tools/bench_ops.py [--batch 8] [--height 384 --width 512] > profiles/rNN_ops_bandwidth.json
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flownet2-tf_amd"))
from src import _hip  # noqa: E402

HBM_PEAK = 8000.0


def timed(fn, rounds, inner):
    ts = []
    for r in range(rounds + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(e0.elapsed_time(e1) / inner)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--inner", type=int, default=10)
    a = ap.parse_args()
    N, H, W = a.batch, a.height, a.width
    dev = _hip.require_device()
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    out = {}

    def rec(name, ms, nbytes, note):
        out[name] = {"ms": round(ms, 5), "algorithmic_MB": round(nbytes / 1e6, 2), "GB_per_s": round(nbytes / ms / 1e6, 1),
                     "frac_of_hbm_peak": round(nbytes / ms / 1e6 / HBM_PEAK, 4), "note": note}

    lib, st = _hip.lib(), _hip.stream_ptr
    P_ = _hip.ptr
    buf = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
    # the C ABI is called directly on preallocated outputs: the Python op wrappers add 10-20 us of host work per
    # call (allocation, argument checks), more than several of these kernels take
    # correlation at the FlowNetC call site: H/8 x W/8 x 256 -> 441 (fp32 op surface)
    h8, w8 = H // 8, W // 8
    fa, fb, co = rnd(N, h8, w8, 256), rnd(N, h8, w8, 256), buf(N, h8, w8, 441)
    ms = timed(lambda: _hip.check(lib.fn2_correlation_f32(P_(fa), P_(fb), P_(co), N, h8, w8, 256, 1, 20, 1, 2, 20, st())),
               a.rounds, a.inner)
    rec("correlation_f32", ms, N * h8 * w8 * (2 * 256 + 441) * 4, "read A, B once + write 441 channels")
    # the same op with a caller-provided workspace: split-fp16 copies of the features + the fp16 matrix-core kernel
    # (what src.correlation.correlation calls for the FlowNetC attribute set)
    need = int(lib.fn2_correlation_workspace_bytes(N, h8, w8, 256, 1, 20, 1, 2, 20))
    if need > 0:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        ms = timed(lambda: _hip.check(lib.fn2_correlation_f32_ws(P_(fa), P_(fb), P_(co), N, h8, w8, 256, 1, 20, 1, 2, 20,
                                                                 P_(ws), need, st())), a.rounds, a.inner)
        rec("correlation_f32_ws", ms, N * h8 * w8 * (2 * 256 + 441) * 4,
            "read A, B once + write 441 channels (+ the two split-fp16 conversion passes, inside the timed call)")
    # the backward ops of SURVEY 8(a) row a12 (used only when a network with these ops is trained)
    gco, gda, gdb = rnd(N, h8, w8, 441), buf(N, h8, w8, 256), buf(N, h8, w8, 256)
    ms = timed(lambda: _hip.check(lib.fn2_correlation_grad_f32(P_(gco), P_(fa), P_(fb), P_(gda), P_(gdb), N, h8, w8, 256,
                                                                 1, 20, 1, 2, 20, st())), max(3, a.rounds // 3), 2)
    rec("correlation_grad_f32", ms, N * h8 * w8 * (4 * 256 + 441) * 4, "read grad, A, B once + write dA, dB")
    # flow_warp at full resolution
    img, flow, wo = rnd(N, H, W, 3), rnd(N, H, W, 2) * 4, buf(N, H, W, 3)
    ms = timed(lambda: _hip.check(lib.fn2_flow_warp_f32(P_(img), P_(flow), P_(wo), N, H, W, 3, st())), a.rounds, a.inner)
    rec("flow_warp_f32", ms, N * H * W * (3 + 2 + 3) * 4, "image + flow + output")
    gwo, gimg, gflow = rnd(N, H, W, 3), buf(N, H, W, 3), buf(N, H, W, 2)
    ms = timed(lambda: _hip.check(lib.fn2_flow_warp_grad_f32(P_(img), P_(flow), P_(gwo), P_(gimg), P_(gflow), N, H, W, 3, st())),
               a.rounds, a.inner)
    rec("flow_warp_grad_f32", ms, N * H * W * (3 + 2 + 3 + 3 + 2) * 4, "image, flow, grad + image_grad (zeroed, atomics), flow_grad")
    # downsample of the ground-truth flow to the coarsest and finest loss scales
    gt = rnd(N, H, W, 2)
    for lvl in (6, 2):
        h, w = H >> lvl, W >> lvl
        do = buf(N, h, w, 2)
        ms = timed(lambda: _hip.check(lib.fn2_downsample_f32(P_(gt), P_(do), N, H, W, 2, h, w, st())), a.rounds, a.inner)
        rec("downsample_to_%dx%d" % (h, w), ms, N * (H * W + h * w) * 2 * 4, "input once + output")
    # resize of predict_flow2 to full resolution
    pf2, ro = rnd(N, H // 4, W // 4, 2), buf(N, H, W, 2)
    ms = timed(lambda: _hip.check(lib.fn2_resize_bilinear_f32(P_(pf2), P_(ro), N, H // 4, W // 4, 2, H, W,
                                                              _hip.C.c_float(20.0), st())), a.rounds, a.inner)
    rec("resize_bilinear_x4", ms, N * (H * W // 16 + H * W) * 2 * 4, "input once + output")
    # augmentation passes (FlyingChairs: crop 7/8 of the width)
    oh, ow = H, W * 7 // 8
    tr = torch.tensor([[1.0, 0.02, 5.0, -0.02, 1.0, 3.0]] * N).to(dev)
    ch = torch.tensor([[1.1, 0.02, 1.05, 0.95, 1.0, 1.05]] * N).to(dev)
    im01, ao, fo = torch.rand(N, H, W, 3, generator=g).to(dev), buf(N, oh, ow, 3), buf(N, oh, ow, 2)
    ms = timed(lambda: _hip.check(lib.fn2_augment_f32(P_(im01), P_(tr), P_(ch), P_(ao), N, H, W, 3, oh, ow, st())),
               a.rounds, a.inner)
    rec("augment_spatial_chromatic", ms, N * (H * W + oh * ow) * 3 * 4, "source once + crop")
    ms = timed(lambda: _hip.check(lib.fn2_flow_augmentation_f32(P_(gt), P_(tr), P_(tr), P_(fo), N, H, W, oh, ow, st())),
               a.rounds, a.inner)
    rec("flow_augmentation", ms, N * (H * W + oh * ow) * 2 * 4, "flow once + crop")
    print(json.dumps({"batch": N, "height": H, "width": W, "hbm_peak_GBps": HBM_PEAK, "ops": out}, indent=1))


if __name__ == "__main__":
    main()
