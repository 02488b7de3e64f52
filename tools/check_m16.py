import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "flownet2-tf_amd"))
import ab_conv
from src import _hip
for name in ("conv2", "conv3_1", "fuse_conv1_1"):
    d, flop, keep = ab_conv.build(name, "f16x2")
    outs = []
    for dbg in ("0", "1024"):
        os.environ["FN2_CONV_DBG"] = dbg
        keep[2].zero_()
        _hip.check(_hip.lib().fn2_conv2d(C.byref(d), _hip.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(keep[2].clone())
    from src import weights as W
    a = W.join_f16x2(outs[0].cpu().numpy().view(np.float16))
    b = W.join_f16x2(outs[1].cpu().numpy().view(np.float16))
    cout = ab_conv.LAYERS[name][5]
    print(name, "max abs diff", float(np.abs(a - b)[..., :cout].max()), "max |a|", float(np.abs(a).max()),
          "first mismatching channel", int(np.argmax(np.abs(a - b).reshape(-1, a.shape[-1]).max(0) > 1e-4)))
