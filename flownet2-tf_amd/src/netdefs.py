"""Layer tables of the six networks (what the reference's model files declare:
src/flownet_s/flownet_s.py:39-104, flownet_c/flownet_c.py:30-107,
flownet_sd/flownet_sd.py:29-103, flownet2/flownet2.py:61-98) as data.

Each entry: (name, kind, k, stride, pad, cin, cout, act).  kind is "conv"
(slim.conv2d on an explicitly padded input) or "deconv" (slim.conv2d_transpose
k=4 s=2 VALID + antipad(1)).  Whether a layer owns a ``biases`` variable is what
the arg_scopes of its model file say -- ``has_bias`` below.  Variable names are
``<scope>/<name>/weights|biases`` with HWIO (conv) / HW-O-I (deconv) layout
 so converted TF checkpoints can be loaded unchanged.
"""

LEAKY, LINEAR = True, False


def _refine(skip_c, interconv):
    """4-level decoder shared by S / C / SD.  skip_c = channels of the skip tensors at levels 5..2."""
    layers = [("predict_flow6", "conv", 3, 1, 1, 1024, 2, LINEAR)]
    cur = 1024
    for lvl, skip, dec in zip((5, 4, 3, 2), skip_c, (512, 256, 128, 64)):
        layers.append((f"deconv{lvl}", "deconv", 4, 2, 1, cur, dec, LEAKY))
        layers.append((f"upsample_flow{lvl + 1}to{lvl}", "deconv", 4, 2, 1, 2, 2, LINEAR))
        cur = skip + dec + 2
        head_in = cur
        if interconv:
            layers.append((f"interconv{lvl}", "conv", 3, 1, 1, cur, dec, LINEAR))
            head_in = dec
        layers.append((f"predict_flow{lvl}", "conv", 3, 1, 1, head_in, 2, LINEAR))
    return layers


def flownet_s_layers(cin=6):
    enc = [("conv1", "conv", 7, 2, 3, cin, 64, LEAKY), ("conv2", "conv", 5, 2, 2, 64, 128, LEAKY),
           ("conv3", "conv", 5, 2, 2, 128, 256, LEAKY), ("conv3_1", "conv", 3, 1, 1, 256, 256, LEAKY),
           ("conv4", "conv", 3, 2, 1, 256, 512, LEAKY), ("conv4_1", "conv", 3, 1, 1, 512, 512, LEAKY),
           ("conv5", "conv", 3, 2, 1, 512, 512, LEAKY), ("conv5_1", "conv", 3, 1, 1, 512, 512, LEAKY),
           ("conv6", "conv", 3, 2, 1, 512, 1024, LEAKY), ("conv6_1", "conv", 3, 1, 1, 1024, 1024, LEAKY)]
    return enc + _refine((512, 512, 256, 128), False)


def flownet_c_layers():
    enc = [("conv1", "conv", 7, 2, 3, 3, 64, LEAKY), ("conv2", "conv", 5, 2, 2, 64, 128, LEAKY),
           ("conv3", "conv", 5, 2, 2, 128, 256, LEAKY), ("conv_redir", "conv", 1, 1, 0, 256, 32, LEAKY),
           ("conv3_1", "conv", 3, 1, 1, 473, 256, LEAKY),
           ("conv4", "conv", 3, 2, 1, 256, 512, LEAKY), ("conv4_1", "conv", 3, 1, 1, 512, 512, LEAKY),
           ("conv5", "conv", 3, 2, 1, 512, 512, LEAKY), ("conv5_1", "conv", 3, 1, 1, 512, 512, LEAKY),
           ("conv6", "conv", 3, 2, 1, 512, 1024, LEAKY), ("conv6_1", "conv", 3, 1, 1, 1024, 1024, LEAKY)]
    return enc + _refine((512, 512, 256, 128), False)


def flownet_sd_layers():
    enc = [("conv0", "conv", 3, 1, 1, 6, 64, LEAKY), ("conv1", "conv", 3, 2, 1, 64, 64, LEAKY),
           ("conv1_1", "conv", 3, 1, 1, 64, 128, LEAKY), ("conv2", "conv", 3, 2, 1, 128, 128, LEAKY),
           ("conv2_1", "conv", 3, 1, 1, 128, 128, LEAKY), ("conv3", "conv", 3, 2, 1, 128, 256, LEAKY),
           ("conv3_1", "conv", 3, 1, 1, 256, 256, LEAKY), ("conv4", "conv", 3, 2, 1, 256, 512, LEAKY),
           ("conv4_1", "conv", 3, 1, 1, 512, 512, LEAKY), ("conv5", "conv", 3, 2, 1, 512, 512, LEAKY),
           ("conv5_1", "conv", 3, 1, 1, 512, 512, LEAKY), ("conv6", "conv", 3, 2, 1, 512, 1024, LEAKY),
           ("conv6_1", "conv", 3, 1, 1, 1024, 1024, LEAKY)]
    return enc + _refine((512, 512, 256, 128), True)


def fusion_layers():
    return [("fuse_conv0", "conv", 3, 1, 1, 11, 64, LEAKY), ("fuse_conv1", "conv", 3, 2, 1, 64, 64, LEAKY),
            ("fuse_conv1_1", "conv", 3, 1, 1, 64, 128, LEAKY), ("fuse_conv2", "conv", 3, 2, 1, 128, 128, LEAKY),
            ("fuse_conv2_1", "conv", 3, 1, 1, 128, 128, LEAKY),
            ("predict_flow2", "conv", 3, 1, 1, 128, 2, LINEAR),
            ("fuse_deconv1", "deconv", 4, 2, 1, 128, 32, LEAKY),
            ("fuse_upsample_flow2to1", "deconv", 4, 2, 1, 2, 2, LINEAR),
            ("fuse_interconv1", "conv", 3, 1, 1, 162, 32, LINEAR),
            ("predict_flow1", "conv", 3, 1, 1, 32, 2, LINEAR),
            ("fuse_deconv0", "deconv", 4, 2, 1, 162, 16, LEAKY),
            ("fuse_upsample_flow1to0", "deconv", 4, 2, 1, 2, 2, LINEAR),
            ("fuse_interconv0", "conv", 3, 1, 1, 82, 16, LINEAR),
            ("predict_flow0", "conv", 3, 1, 1, 16, 2, LINEAR)]


def has_bias(model, name, kind, no_deconv_biases=True):
    """Does the reference graph create ``<scope>/<name>/biases``?  slim's default is
    biases_initializer=zeros (a variable exists) unless an arg_scope or the call says None:

      * every slim.conv2d of S / C / SD / fusion: bias (no model file overrides it);
      * slim.conv2d_transpose inside the refinement scopes of S / C / SD: biases_initializer=None
        (flownet_s.py:53, flownet_c.py:58, flownet_sd.py:44) -> deconvN and upsample_flowXtoY have none;
      * the FlowNet2 fusion net opens NO such scope (flownet2.py:50-57): fuse_deconv1/0 and
        fuse_upsample_flow2to1/1to0 (flownet2.py:66-89) DO carry biases, and the Caffe converter writes them
        (scripts/caffe/convert_caffe_weights_to_npy.py:454-459, :492-496);
      * FlowNetS_interp (flownet_s_interp.py:78-126): predict_flowN and deconvN take
        ``biases_initializer = None if no_deconv_biases else zeros``; upsample_flowXtoY always None."""
    fused = name.startswith("fuse_")
    if kind == "conv":
        if model == "FlowNetS_interp" and name.startswith("predict_flow"):
            return not no_deconv_biases
        return True
    if fused:
        return True
    if model == "FlowNetS_interp" and name.startswith("deconv"):
        return not no_deconv_biases
    return False


def model_scopes(model):
    """[(variable scope, layer table)] of a model; scope nesting as the reference's
    tf.variable_scope calls produce it (flownet_cs.py:16, flownet_css.py:16, flownet2.py:20)."""
    if model in ("FlowNetS", "FlowNetS_interp"):  # the interpolation net keeps the scope (flownet_s_interp.py:23)
        return [("FlowNetS", flownet_s_layers(6))]
    if model == "FlowNetC":
        return [("FlowNetC", flownet_c_layers())]
    if model == "FlowNetSD":
        return [("FlowNetSD", flownet_sd_layers())]
    if model == "FlowNetCS":
        return [("FlowNetCS/FlowNetC", flownet_c_layers()), ("FlowNetCS/FlowNetS", flownet_s_layers(12))]
    if model == "FlowNetCSS":
        return [("FlowNetCSS/FlowNetCS/FlowNetC", flownet_c_layers()),
                ("FlowNetCSS/FlowNetCS/FlowNetS", flownet_s_layers(12)),
                ("FlowNetCSS/FlowNetS", flownet_s_layers(12))]
    if model == "FlowNet2":
        return [("FlowNet2/FlowNetCSS/FlowNetCS/FlowNetC", flownet_c_layers()),
                ("FlowNet2/FlowNetCSS/FlowNetCS/FlowNetS", flownet_s_layers(12)),
                ("FlowNet2/FlowNetCSS/FlowNetS", flownet_s_layers(12)),
                ("FlowNet2/FlowNetSD", flownet_sd_layers()),
                ("FlowNet2", fusion_layers())]
    raise ValueError("unknown model %r" % model)


MODELS = ("FlowNetS", "FlowNetC", "FlowNetSD", "FlowNetCS", "FlowNetCSS", "FlowNet2", "FlowNetS_interp")
