"""Run ONE conv layer of tools/ab_conv.py's table repeatedly (for rocprofv3 --pmc passes):
   python tools/diag/pmc_layer.py conv4_1_b4 [--fragments] [--reps 20]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "flownet2-tf_amd"), os.path.join(ROOT, "tools")]
import torch
from src import _hip, weights as W
import ab_conv
name = sys.argv[1]
frag = "--fragments" in sys.argv
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 20
d, flop, keep = ab_conv.build(name, "f16x2")
if frag:
    wf = W.to_fragment_order(keep[1])
    d.wgt, d.wgt_layout = wf.data_ptr(), 2
buf = C.create_string_buffer(256)
_hip.lib().fn2_conv2d_kernel_name(C.byref(d), buf, 256)
print(name, buf.value.decode(), "GFLOP %.2f" % (flop / 1e9))
s = _hip.stream_ptr()
for _ in range(reps):
    _hip.check(_hip.lib().fn2_conv2d(C.byref(d), s))
torch.cuda.synchronize()
