"""``src.downsample`` -- drop-in for /root/reference src/downsample.py:7-8:
``downsample(tensor, size)`` with ``size = [h, w]`` -> ``N x h x w x C`` float32,
the NaN-aware weighted-area downsample of downsample_kernel_gpu.cu.cc:35-76 on
libflownet2_hip.so (fn2_downsample_f32).  No gradient (the reference registers none)."""
import torch

from . import _hip


def downsample(tensor, size):
    size = [int(s) for s in size]
    if len(size) != 2:
        raise ValueError("size must have 2 elements")  # downsample_kernel.cc:19
    x, kind = _hip.to_device_f32(tensor)
    if x.dim() != 4:
        raise ValueError("Input images must have rank 4")  # downsample_kernel.cc:25
    n, h, w, c = x.shape
    out = torch.empty((n, size[0], size[1], c), dtype=torch.float32, device=x.device)
    _hip.check(_hip.lib().fn2_downsample_f32(_hip.ptr(x), _hip.ptr(out), n, h, w, c, size[0], size[1],
                                             _hip.stream_ptr()))
    return _hip.from_device(out, kind)
