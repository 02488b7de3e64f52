"""``src.flow_warp`` -- drop-in for /root/reference src/flow_warp.py:7-15:
``flow_warp(image, flow)`` -> tensor shaped like ``image`` (NHWC float32), backward
bilinear warp with the reference kernel's exact in-range / clamp rules
(flow_warp.cu.cc:44-95), on libflownet2_hip.so (fn2_flow_warp_f32 /
fn2_flow_warp_grad_f32).  torch tensors or numpy arrays in, same kind out."""
import torch

from . import _hip


def _check(image, flow):
    if image.dim() != 4:
        raise ValueError("Input images must have rank 4")  # flow_warp.cc:22
    if flow.dim() != 4:
        raise ValueError("Input flow must have rank 4")  # flow_warp.cc:23
    if image.shape[:3] != flow.shape[:3]:
        raise ValueError("Input images and flows must have the same batch, height and width")  # :24-29
    if flow.shape[3] != 2:
        raise ValueError("Input flow must have 2 channels")  # :30


def _forward(image, flow):
    n, h, w, c = image.shape
    out = torch.empty_like(image)
    _hip.check(_hip.lib().fn2_flow_warp_f32(_hip.ptr(image), _hip.ptr(flow), _hip.ptr(out), n, h, w, c,
                                            _hip.stream_ptr()))
    return out


class _FlowWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, flow):
        ctx.save_for_backward(image, flow)
        return _forward(image, flow)

    @staticmethod
    def backward(ctx, grad):
        image, flow = ctx.saved_tensors
        n, h, w, c = image.shape
        grad = grad.contiguous().float()
        dimg, dflow = torch.empty_like(image), torch.empty_like(flow)
        _hip.check(_hip.lib().fn2_flow_warp_grad_f32(_hip.ptr(image), _hip.ptr(flow), _hip.ptr(grad),
                                                     _hip.ptr(dimg), _hip.ptr(dflow), n, h, w, c,
                                                     _hip.stream_ptr()))
        return dimg, dflow


def flow_warp(image, flow):
    needs_grad = any(isinstance(t, torch.Tensor) and t.requires_grad for t in (image, flow))
    if needs_grad:
        if not (image.is_cuda and flow.is_cuda):
            raise RuntimeError("flow_warp gradients need ROCm tensors (no CPU path)")
        _check(image, flow)
        return _FlowWarp.apply(image.contiguous().float(), flow.contiguous().float())
    img, kind = _hip.to_device_f32(image)
    fl, _ = _hip.to_device_f32(flow)
    _check(img, fl)
    return _hip.from_device(_forward(img, fl), kind)
