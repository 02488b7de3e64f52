"""fn2_flow_head5 at the fusion net's two call sites, with the ablation bits of FN2_H5_DBG (2: K loop only, 8: no gather,
4: plain block order)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "flownet2-tf_amd"))
from src import _hip, weights as W
lib = _hip.lib()
rng = np.random.default_rng(0)
for (N, H, Wd, cin) in ((4, 384, 512, 82), (4, 192, 256, 162)):
    cs = (cin + 31) // 32 * 32
    x = torch.from_numpy(W.split_f16x2(rng.standard_normal((N, H, Wd, cs)).astype(np.float32)).view(np.float32)).cuda()
    w5 = rng.standard_normal((5, 5, cin, 2)).astype(np.float32) * 0.02
    plan = _hip.conv_plan(3, cs, 50)
    packed, cin_pad, cout_pad, kpad = W.pack_conv(np.ascontiguousarray(w5.transpose(2, 0, 1, 3)).reshape(1, 1, cin, 50), plan.cout_tile, plan.kstep_elems, cs, plan.layout)
    wdev = W.packed_to_device(packed, plan.wgt_dtype, "cuda")
    bd = torch.zeros(2, device="cuda")
    pf = torch.zeros((N, H, Wd, 2), device="cuda")
    v = _hip.view(x, cin, 0, 3)
    groups = (cin + 7) // 8
    wcd = torch.from_numpy(rng.standard_normal((9, 25, groups * 8, 2)).astype(np.float32) * 0.02).cuda()
    bcd = torch.zeros((9, 2), device="cuda")
    # tile form: 0 / 4 / 8 / 2 as above; strip form (FN2_H5_STRIP=1): 16 no ring share, 48 + no gather, 112 + no MFMA
    for strip, dbg in (("0", "0"), ("0", "8"), ("0", "2"), ("1", "0"), ("1", "16"), ("1", "48"), ("1", "112")):
        os.environ["FN2_H5_DBG"] = dbg
        os.environ["FN2_H5_STRIP"] = strip
        run = lambda: _hip.check(lib.fn2_flow_head5(C.byref(v), wdev.data_ptr(), cin_pad, kpad, C.c_float(1.0), bd.data_ptr(), pf.data_ptr(), 1, wcd.data_ptr(), bcd.data_ptr(), _hip.stream_ptr()))
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        print("%dx%dx%d cin %d  strip=%s dbg=%s  %.1f us" % (N, H, Wd, cin, strip, dbg, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
