"""corr3 time vs batch: is the kernel's time quantised in rounds of one block per CU?"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "flownet2-tf_amd"))
from src import _hip, weights as W
lib = _hip.lib()
args = sys.argv[1:]
if "--lib" in args:  # an alternative build of the library (tools/build_variant.sh)
    path = args[args.index("--lib") + 1]
    args = [a for a in args if a not in ("--lib", path)]
    lib = C.CDLL(os.path.abspath(path))
    for name, (res, at) in _hip.PROTOTYPES.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, at
    print("library:", path)
rng = np.random.default_rng(0)
for batch in ([int(v) for v in args] or (1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 16)):
    shp = (batch, 48, 64, 256)
    fa = torch.from_numpy(W.split_f16x2(rng.standard_normal(shp).astype(np.float32)).view(np.float32)).cuda()
    fb = torch.from_numpy(W.split_f16x2(rng.standard_normal(shp).astype(np.float32)).view(np.float32)).cuda()
    out = torch.zeros((batch, 48, 64, 480), dtype=torch.float32, device="cuda")
    va, vb, vo = _hip.view(fa, 256, 0, 3), _hip.view(fb, 256, 0, 3), _hip.view(out, 441, 32, 3)
    run = lambda: _hip.check(lib.fn2_correlation_fused(C.byref(va), C.byref(vb), C.byref(vo), 20, 2, 1, _hip.stream_ptr()))
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    ref = getattr(sys.modules[__name__], "_ref_%d" % batch, None)
    print("batch %2d: %4d blocks (%.2f per CU)  %6.1f us  %.2f us per block-round-equivalent" % (batch, 72 * batch, 72 * batch / 256, us, us / np.ceil(72 * batch / 256)))
