"""ctypes binding of libflownet2_hip.so (include/flownet2_hip.h).

There is no CPU fallback: every op in this package needs the HIP library and a
ROCm device, and says so loudly when either is missing.
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libflownet2_hip.so")

FN2_F32, FN2_BF16, FN2_F16, FN2_F16X2 = 0, 1, 2, 3
ACT_NONE, ACT_LEAKY = 0, 1
OK = 0
ERR_INVALID_ARGUMENT, ERR_UNSUPPORTED, ERR_HIP = -1, -2, -3


class Fn2Tensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32),
                ("w", C.c_int32), ("c", C.c_int32), ("cs", C.c_int32), ("c0", C.c_int32)]


class Fn2ConvPlan(C.Structure):
    _fields_ = [("layout", C.c_int32), ("cout_tile", C.c_int32), ("kstep_elems", C.c_int32),
                ("wgt_dtype", C.c_int32)]


class Fn2ConvDesc(C.Structure):
    _fields_ = [("inp", Fn2Tensor), ("out", Fn2Tensor), ("wgt", C.c_void_p), ("bias", C.c_void_p),
                ("kind", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32),
                ("pad", C.c_int32), ("act", C.c_int32), ("cin_pad", C.c_int32), ("cout_pad", C.c_int32),
                ("kpad", C.c_int32), ("wgt_layout", C.c_int32), ("accumulate", C.c_int32), ("out_scale", C.c_float), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("act_grad_y", C.c_void_p), ("act_grad_c0", C.c_int32), ("act_grad_c1", C.c_int32),
                ("up_src", C.c_void_p), ("up_w", C.c_void_p), ("up_bias", C.c_void_p), ("up_c0", C.c_int32),
                ("head", C.c_void_p), ("raw_partials", C.c_int32)]


class Fn2BwdwDesc(C.Structure):
    _fields_ = [("x", Fn2Tensor), ("dy", Fn2Tensor), ("dw", C.c_void_p), ("kind", C.c_int32), ("kh", C.c_int32),
                ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("cin_pad", C.c_int32),
                ("cout_pad", C.c_int32), ("kpad", C.c_int32), ("wgt_layout", C.c_int32), ("db", C.c_void_p)]


_i, _p, _f = C.c_int, C.c_void_p, C.c_float
_ip = C.POINTER(C.c_int)
_tp = C.POINTER(Fn2Tensor)

# name -> (restype, argtypes); every symbol include/flownet2_hip.h declares
PROTOTYPES = {
    "fn2_last_error": (C.c_char_p, []),
    "fn2_version": (_i, []),
    "fn2_crc32c": (C.c_uint32, [_p, C.c_int64, C.c_uint32]),
    "fn2_device_info": (_i, [C.c_char_p, _i, _ip]),
    "fn2_correlation_out_shape": (_i, [_i] * 7 + [_ip] * 3),
    "fn2_correlation_f32": (_i, [_p, _p, _p] + [_i] * 9 + [_p]),
    "fn2_correlation_workspace_bytes": (C.c_int64, [_i] * 9),
    "fn2_correlation_f32_ws": (_i, [_p, _p, _p] + [_i] * 9 + [_p, C.c_int64, _p]),
    "fn2_correlation_grad_f32": (_i, [_p] * 5 + [_i] * 9 + [_p]),
    "fn2_flow_warp_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "fn2_flow_warp_grad_f32": (_i, [_p] * 5 + [_i] * 4 + [_p]),
    "fn2_downsample_f32": (_i, [_p, _p] + [_i] * 6 + [_p]),
    "fn2_downsample_scaled_f32": (_i, [_p, _f, _p] + [_i] * 6 + [_p]),
    "fn2_fill_zero": (_i, [_p, C.c_int64, _p]),
    "fn2_add_f32": (_i, [_p, _p, C.c_int64, _p]),
    "fn2_slice_copy_f32": (_i, [_tp, _p, _p]),
    "fn2_resize_bilinear_f32": (_i, [_p, _p] + [_i] * 6 + [_f, _p]),
    "fn2_conv2d_plan": (_i, [_i, _i, _i, C.POINTER(Fn2ConvPlan)]),
    "fn2_conv2d_workspace_bytes": (C.c_int64, [C.POINTER(Fn2ConvDesc)]),
    "fn2_conv2d": (_i, [C.POINTER(Fn2ConvDesc), _p]),
    "fn2_conv2d_kernel_name": (_i, [C.POINTER(Fn2ConvDesc), C.c_char_p, _i]),
    "fn2_conv2d_splits": (_i, [C.POINTER(Fn2ConvDesc)]),
    "fn2_augment_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "fn2_flow_augmentation_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "fn2_flow_head_gather": (_i, [_p, _i, _p, _p, _i, _i, _i, _p]),
    "fn2_upsample_flow": (_i, [_p, _p, _p, _tp, _i, _i, _i, _p]),
    "fn2_flow_head_tail": (_i, [_p, _i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _tp, _p]),
    "fn2_flow_head_tail_slabs": (_i, [_p, _i, _i, C.c_int64, _f, _i, _p, _p, _i, _i, _i, _i, _p, _p, _tp, _p]),
    "fn2_flow_head_ring": (_i, [_tp, _p, _p, _p, _p]),
    "fn2_flow_head5": (_i, [_tp, _p, _i, _i, _f, _p, _p, _i, _p, _p, _p]),
    "fn2_u8_to_f32_lut": (_i, [_p, _p, _p, C.c_int64, _p]),
    "fn2_pack_pair": (_i, [_p, _p, _tp, _i, _p]),
    "fn2_pack_image": (_i, [_p, _i, _tp, _i, _i, _p]),
    "fn2_pack_image_s2d": (_i, [_p, _i, _i, _i, _tp, _i, _i, _p]),
    "fn2_correlation_fused": (_i, [_tp, _tp, _tp, _i, _i, _i, _p]),
    "fn2_stack_input": (_i, [_p, _p, _p, _tp, _i, _p]),
    "fn2_fusion_input": (_i, [_p, _p, _p, _p, _tp, _i, _p]),
    "fn2_stack_input_pf": (_i, [_p, _p, _p, _i, _i, _f, _p, _tp, _i, _p]),
    "fn2_fusion_input_pf": (_i, [_p, _p, _p, _f, _p, _p, _f, _p, _i, _i, _tp, _i, _p]),
    "fn2_epe_loss_grad": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _f, _p]),
    "fn2_epe_loss_grad_weighted": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _p]),
    "fn2_leaky_bwd": (_i, [_tp, _tp, _p, _p]),
    "fn2_bias_grad": (_i, [_tp, _p, _p]),
    "fn2_gather_f32": (_i, [_p, _p, _p, C.c_int64, _p]),
    "fn2_to_f16x2": (_i, [_p, _p, _p, C.c_int64, _f, _p]),
    "fn2_to_f16x2_frag": (_i, [_p, _p, _p, C.c_int64, _f, _i, _p]),
    "fn2_adam_step": (_i, [_p, _p, _p, _p, C.c_int64, _f, _f, _f, _f, _i, _f, _f, _p]),
    "fn2_adam_step_multi": (_i, [_p, _p, _p, _i, _f, _f, _f, _f, _i, _f, _p]),
    "fn2_adam_step_multi_dev": (_i, [_p, _p, _p, _i, _p, _p]),
    "fn2_upsample_flow_bwd": (_i, [_tp, _p, _p, _p, _p, _i, _p]),
    "fn2_head_bwd_filter": (_i, [_tp, _p, _p, _i, _i, _p]),
    "fn2_head_g18": (_i, [_p, _tp, _p]),
    "fn2_head_bwd_data": (_i, [_p, _p, _tp, _i, _i, _p]),
    "fn2_conv2d_bwd_filter": (_i, [C.POINTER(Fn2BwdwDesc), _p]),
    "fn2_capture_begin": (_i, [_p]),
    "fn2_capture_end": (_i, [_p, C.POINTER(C.c_void_p)]),
    "fn2_graph_launch": (_i, [_p, _p]),
    "fn2_graph_destroy": (_i, [_p]),
}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                "libflownet2_hip.so not found at %s -- build it with `make -C flownet2-tf_amd/csrc` "
                "(or __graft_entry__.build()); this package has no CPU fallback" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    """Map the C status to the exceptions the reference's Python surface raises:
    errors::InvalidArgument -> ValueError."""
    if rc == OK:
        return
    msg = lib().fn2_last_error().decode("utf-8", "replace")
    if rc == ERR_INVALID_ARGUMENT:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError("HIP error: " + msg)


def require_device():
    if not torch.cuda.is_available():
        raise RuntimeError("no ROCm device visible: the FlowNet2 ops run on MI355X only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr())


def dtype_code(t):
    if t.dtype == torch.float32:
        return FN2_F32
    if t.dtype == torch.bfloat16:
        return FN2_BF16
    if t.dtype == torch.float16:
        return FN2_F16
    raise ValueError("unsupported dtype %s" % t.dtype)


def view(buf, c=None, c0=0, code=None):
    """fn2_tensor over channels [c0, c0+c) of a dense NHWC torch buffer.  `code` overrides the dtype
    derived from the torch dtype (split-fp16 buffers live in float32 containers)."""
    n, h, w, cs = buf.shape
    assert buf.is_contiguous()
    return Fn2Tensor(buf.data_ptr(), dtype_code(buf) if code is None else code, n, h, w,
                     cs - c0 if c is None else c, cs, c0)


def conv_plan(in_code, cin_pad, cout):
    p = Fn2ConvPlan()
    check(lib().fn2_conv2d_plan(in_code, cin_pad, cout, C.byref(p)))
    return p


def to_device_f32(x):
    """Accept torch.Tensor (any device) or numpy.ndarray; return (fp32 contiguous ROCm tensor, kind)."""
    dev = require_device()
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev), "numpy"
    if isinstance(x, torch.Tensor):
        kind = "cuda" if x.is_cuda else "cpu"
        return x.detach().to(device=dev, dtype=torch.float32).contiguous(), kind
    raise TypeError("expected torch.Tensor or numpy.ndarray, got %r" % type(x))


def from_device(t, kind):
    if kind == "numpy":
        return t.cpu().numpy()
    if kind == "cpu":
        return t.cpu()
    return t
