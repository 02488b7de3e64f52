"""``src.correlation`` -- drop-in for the reference module of the same name
(/root/reference: src/correlation.py:7-35).  Same function, same positional
arguments, NHWC float32 in, ``N x oh x ow x D`` out; the TF op library is
replaced by libflownet2_hip.so (fn2_correlation_f32 / fn2_correlation_grad_f32).

Accepts torch tensors (ROCm or CPU) or numpy arrays and returns the same kind.
Gradients: where the reference registers ``@tf.RegisterGradient("Correlation")``
(correlation.py:17-35) this registers a torch.autograd.Function.
"""
import ctypes as C

import torch

from . import _hip


def _out_shape(h, w, k, md, s1, s2, pad):
    oh, ow, oc = C.c_int(), C.c_int(), C.c_int()
    _hip.check(_hip.lib().fn2_correlation_out_shape(h, w, k, md, s1, s2, pad, C.byref(oh), C.byref(ow),
                                                    C.byref(oc)))
    return oh.value, ow.value, oc.value


def _check_inputs(a, b):
    if a.dim() != 4:
        raise ValueError("input_a must have rank 4")  # correlation_kernel.cc:31
    if b.dim() != 4:
        raise ValueError("input_b must have rank 4")  # correlation_kernel.cc:32
    if a.shape != b.shape:
        raise ValueError("input_a and input_b must have the same shape")  # correlation_op.cc:17


def _forward(a, b, k, md, s1, s2, pad):
    n, h, w, c = a.shape
    oh, ow, oc = _out_shape(h, w, k, md, s1, s2, pad)
    out = torch.empty((n, oh, ow, oc), dtype=torch.float32, device=a.device)
    lib = _hip.lib()
    need = int(lib.fn2_correlation_workspace_bytes(n, h, w, c, k, md, s1, s2, pad))
    if need > 0:  # the FlowNetC attribute set: split-fp16 copies of the features + the matrix-core kernel
        ws = torch.empty(need, dtype=torch.uint8, device=a.device)
        _hip.check(lib.fn2_correlation_f32_ws(_hip.ptr(a), _hip.ptr(b), _hip.ptr(out), n, h, w, c, k, md, s1, s2, pad,
                                              _hip.ptr(ws), need, _hip.stream_ptr()))
        return out
    _hip.check(lib.fn2_correlation_f32(_hip.ptr(a), _hip.ptr(b), _hip.ptr(out), n, h, w, c,
                                       k, md, s1, s2, pad, _hip.stream_ptr()))
    return out


class _Correlation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, k, md, s1, s2, pad):
        ctx.save_for_backward(a, b)
        ctx.attrs = (k, md, s1, s2, pad)
        return _forward(a, b, k, md, s1, s2, pad)

    @staticmethod
    def backward(ctx, grad):
        a, b = ctx.saved_tensors
        k, md, s1, s2, pad = ctx.attrs
        n, h, w, c = a.shape
        grad = grad.contiguous().float()
        da, db = torch.empty_like(a), torch.empty_like(b)
        _hip.check(_hip.lib().fn2_correlation_grad_f32(_hip.ptr(grad), _hip.ptr(a), _hip.ptr(b), _hip.ptr(da),
                                                       _hip.ptr(db), n, h, w, c, k, md, s1, s2, pad,
                                                       _hip.stream_ptr()))
        return da, db, None, None, None, None, None


def correlation(input_a, input_b, kernel_size, max_displacement, stride_1, stride_2, padding):
    if int(kernel_size) % 2 == 0:
        raise ValueError("kernel_size must be odd")  # correlation_kernel.cc:23
    needs_grad = any(isinstance(t, torch.Tensor) and t.requires_grad for t in (input_a, input_b))
    if needs_grad:
        if not (input_a.is_cuda and input_b.is_cuda):
            raise RuntimeError("correlation gradients need ROCm tensors (no CPU path)")
        _check_inputs(input_a, input_b)
        return _Correlation.apply(input_a.contiguous().float(), input_b.contiguous().float(), int(kernel_size),
                                  int(max_displacement), int(stride_1), int(stride_2), int(padding))
    a, kind = _hip.to_device_f32(input_a)
    b, _ = _hip.to_device_f32(input_b)
    _check_inputs(a, b)
    out = _forward(a, b, int(kernel_size), int(max_displacement), int(stride_1), int(stride_2), int(padding))
    return _hip.from_device(out, kind)
