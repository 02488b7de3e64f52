#!/usr/bin/env python
"""Timing of fn2_correlation_fused at the FlowNetC call-site size (split fp16 in / out, LeakyReLU, concat slice), with the
ablation bits of corr3.hip (FN2_CORR3_DBG: 1 no DMA, 2 no MFMA, 4 no LDS-tile writes, 8 no global stores) and corr2.

  python tools/ab_corr.py [--batch 8]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=48)
    ap.add_argument("--width", type=int, default=64)
    a = ap.parse_args()
    lib = _hip.lib()
    rng = np.random.default_rng(0)
    shp = (a.batch, a.height, a.width, 256)
    fa = torch.from_numpy(W.split_f16x2(rng.standard_normal(shp).astype(np.float32)).view(np.float32)).cuda()
    fb = torch.from_numpy(W.split_f16x2(rng.standard_normal(shp).astype(np.float32)).view(np.float32)).cuda()
    out = torch.zeros((a.batch, a.height, a.width, 480), dtype=torch.float32, device="cuda")
    va, vb, vo = _hip.view(fa, 256, 0, 3), _hip.view(fb, 256, 0, 3), _hip.view(out, 441, 32, 3)

    def run():
        _hip.check(lib.fn2_correlation_fused(C.byref(va), C.byref(vb), C.byref(vo), 20, 2, 1, _hip.stream_ptr()))

    def timed(n=50):
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    algo = a.batch * a.height * a.width * (2 * 256 + 441) * 4
    for name, env in (("corr2", {"FN2_CORR3": "0"}), ("corr3", {}), ("corr3 no DMA", {"FN2_CORR3_DBG": "1"}),
                      ("corr3 no MFMA", {"FN2_CORR3_DBG": "2"}), ("corr3 no tile writes", {"FN2_CORR3_DBG": "4"}),
                      ("corr3 no stores", {"FN2_CORR3_DBG": "8"}), ("corr3 DMA only", {"FN2_CORR3_DBG": "14"}),
                      ("corr3 nothing", {"FN2_CORR3_DBG": "15"})):
        for k in ("FN2_CORR3", "FN2_CORR3_DBG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        us = timed()
        print("%-22s %7.1f us   %6.2f TB/s of algorithmic bytes" % (name, us, algo / us / 1e6))


if __name__ == "__main__":
    main()
