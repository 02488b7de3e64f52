"""Oracle (test infrastructure): the reference's model graphs, forward + loss,
in NumPy.  PARITY UNPINNED at the TF/cuDNN boundary (see oracle/nn.py).

``weights`` is a dict keyed by the reference's variable names
(``<scope>/<layer>/weights`` HWIO for conv, HW-O-I for conv-transpose,
``<scope>/<layer>/biases``).  ``inputs`` is the dict the
reference's ``model()`` takes: ``input_a``/``input_b`` NHWC in [0, 1].
Activations are float64 (exact-arithmetic meaning of the graph); the three
custom ops run in float32 exactly as the reference kernels do.
"""
import numpy as np

from . import nn, ops

F64 = np.float64
LOSS_WEIGHTS = (0.32, 0.08, 0.02, 0.01, 0.005)  # flownet_s.py:158


class _Scope:
    def __init__(self, weights, scope):
        self.w, self.s = weights, scope

    def conv(self, x, name, stride=1, padding=1, act=True):
        # slim.conv2d under the arg_scope of flownet_s.py:26-37: bias present,
        # LeakyReLU unless activation_fn=None
        w = self.w[f"{self.s}/{name}/weights"]
        key = f"{self.s}/{name}/biases"
        # every slim.conv2d of the model files has its bias variable; only FlowNetS_interp's heads may be built
        # with biases_initializer=None (flownet_s_interp.py:78-126) -- then the entry is absent
        b = self.w[key] if (key in self.w or not name.startswith("predict_flow")) else None
        return nn.conv2d(x, w, b, stride=stride, padding=padding,
                         activation=nn.leaky_relu if act else None)

    def deconv(self, x, name, act=True, bias=False):
        # antipad(slim.conv2d_transpose(.., 4, stride=2)).  bias=False: the call sits inside a
        # biases_initializer=None scope (flownet_s.py:53, flownet_c.py:58, flownet_sd.py:44) -- a
        # ``biases`` entry of the checkpoint, if any, is not a graph variable and is not read.
        # bias=True: slim's default zeros-initialised bias variable exists and MUST be in the checkpoint
        # (the FlowNet2 fusion net, flownet2.py:50-89; FlowNetS_interp's deconvN with no_deconv_biases=False).
        w = self.w[f"{self.s}/{name}/weights"]
        b = self.w[f"{self.s}/{name}/biases"] if bias else None
        return nn.conv2d_transpose(x, w, stride=2, crop=1, bias=b,
                                   activation=nn.leaky_relu if act else None)


def _refine(sc, feats, interconv=False, deconv_bias=False):
    """The 4-level refinement decoder shared by S, C and SD
    (flownet_s.py:52-104; flownet_sd.py:45-103 adds interconvN).
    ``feats`` = (conv6_1, conv5_1, conv4_1, conv3_1, skip2).  deconv_bias: FlowNetS_interp built with
    no_deconv_biases=False gives deconvN (not upsample_flowXtoY) a bias (flownet_s_interp.py:84-126)."""
    top, skips = feats[0], feats[1:]
    preds = {}
    pf = sc.conv(top, "predict_flow6", act=False)
    preds["predict_flow6"] = pf
    cur = top
    for lvl, skip in zip((5, 4, 3, 2), skips):
        dec = sc.deconv(cur, f"deconv{lvl}", bias=deconv_bias)
        up = sc.deconv(pf, f"upsample_flow{lvl + 1}to{lvl}", act=False)
        cur = np.concatenate([skip, dec, up], axis=3)  # [skip | deconv | up], :64
        head_in = sc.conv(cur, f"interconv{lvl}", act=False) if interconv else cur
        pf = sc.conv(head_in, f"predict_flow{lvl}", act=False)
        preds[f"predict_flow{lvl}"] = pf
    return preds


def _finish(preds, height, width, scale):
    flow = preds["predict_flow2"] * scale
    preds["flow"] = nn.resize_bilinear_align_corners(flow, (height, width))
    return preds


def flownet_s(weights, inputs, scope="FlowNetS", deconv_bias=False):
    """FlowNetS.model (flownet_s.py:14-120)."""
    a = np.asarray(inputs["input_a"], F64)
    _, H, W, _ = a.shape
    if "warped" in inputs and "flow" in inputs and "brightness_error" in inputs:  # :18-24
        x = np.concatenate([a, inputs["input_b"], inputs["warped"], inputs["flow"],
                            inputs["brightness_error"]], axis=3).astype(F64)
    else:
        x = np.concatenate([a, inputs["input_b"]], axis=3).astype(F64)
    sc = _Scope(weights, scope)
    c1 = sc.conv(x, "conv1", 2, 3)
    c2 = sc.conv(c1, "conv2", 2, 2)
    c3 = sc.conv(c2, "conv3", 2, 2)
    c3_1 = sc.conv(c3, "conv3_1")
    c4_1 = sc.conv(sc.conv(c3_1, "conv4", 2), "conv4_1")
    c5_1 = sc.conv(sc.conv(c4_1, "conv5", 2), "conv5_1")
    c6_1 = sc.conv(sc.conv(c5_1, "conv6", 2), "conv6_1")
    preds = _refine(sc, (c6_1, c5_1, c4_1, c3_1, c2), deconv_bias=deconv_bias)
    return _finish(preds, H, W, 20.0)  # :107-111


def flownet_s_interp(weights, inputs, scope="FlowNetS", no_deconv_biases=None):
    """FlowNetS_interp.model (flownet_s_interp/flownet_s_interp.py:21-156): the FlowNetS tower on
    [input_a | 0.05 * sparse_flow | matches_a] (:34-38), variable scope 'FlowNetS' (:23); with the class default
    no_deconv_biases=True neither the predict_flow layers nor deconvN have biases, with False both have
    (:78-126); upsample_flowXtoY never.  None: as the weights say (predict_flow6/biases present or not)."""
    if no_deconv_biases is None:
        no_deconv_biases = f"{scope}/predict_flow6/biases" not in weights
    if no_deconv_biases:  # the heads' `biases` of a plain FlowNetS checkpoint are not variables of this graph
        weights = {k: v for k, v in weights.items()
                   if not (k.startswith(scope + "/predict_flow") and k.endswith("/biases"))}
    m = np.asarray(inputs["matches_a"], F64)
    if m.ndim == 3:
        m = m[..., None]
    second = np.concatenate([np.asarray(inputs["sparse_flow"], F64) * 0.05, m], axis=3)
    return flownet_s(weights, {"input_a": inputs["input_a"], "input_b": second}, scope,
                     deconv_bias=not no_deconv_biases)


def flownet_c(weights, inputs, scope="FlowNetC"):
    """FlowNetC.model (flownet_c.py:15-125)."""
    a = np.asarray(inputs["input_a"], F64)
    b = np.asarray(inputs["input_b"], F64)
    _, H, W, _ = a.shape
    sc = _Scope(weights, scope)

    def tower(x):  # shared weights via reuse=True, :30-37
        t1 = sc.conv(x, "conv1", 2, 3)
        t2 = sc.conv(t1, "conv2", 2, 2)
        return t2, sc.conv(t2, "conv3", 2, 2)

    a2, a3 = tower(a)
    _, b3 = tower(b)
    cc = ops.correlation(a3.astype(np.float32), b3.astype(np.float32), 1, 20, 1, 2, 20)  # :40
    cc = nn.leaky_relu(cc.astype(F64))
    redir = sc.conv(a3, "conv_redir", 1, 0)  # 1x1, no pad, :44
    net = np.concatenate([redir, cc], axis=3)  # :46
    c3_1 = sc.conv(net, "conv3_1")
    c4_1 = sc.conv(sc.conv(c3_1, "conv4", 2), "conv4_1")
    c5_1 = sc.conv(sc.conv(c4_1, "conv5", 2), "conv5_1")
    c6_1 = sc.conv(sc.conv(c5_1, "conv6", 2), "conv6_1")
    preds = _refine(sc, (c6_1, c5_1, c4_1, c3_1, a2))  # skip = conv_a_2, :105
    return _finish(preds, H, W, 20.0)


def flownet_sd(weights, inputs, scope="FlowNetSD"):
    """FlowNetSD.model (flownet_sd.py:14-119)."""
    a = np.asarray(inputs["input_a"], F64)
    _, H, W, _ = a.shape
    x = np.concatenate([a, inputs["input_b"]], axis=3).astype(F64)
    sc = _Scope(weights, scope)
    c0 = sc.conv(x, "conv0")
    c1_1 = sc.conv(sc.conv(c0, "conv1", 2), "conv1_1")
    c2 = sc.conv(c1_1, "conv2", 2)
    c2_1 = sc.conv(c2, "conv2_1")
    c3_1 = sc.conv(sc.conv(c2_1, "conv3", 2), "conv3_1")
    c4_1 = sc.conv(sc.conv(c3_1, "conv4", 2), "conv4_1")
    c5_1 = sc.conv(sc.conv(c4_1, "conv5", 2), "conv5_1")
    c6_1 = sc.conv(sc.conv(c5_1, "conv6", 2), "conv6_1")
    preds = _refine(sc, (c6_1, c5_1, c4_1, c3_1, c2), interconv=True)  # skip = conv2, :97
    return _finish(preds, H, W, 0.05)  # :106


def _stack_inputs(inputs, flow):
    """warp + brightness error + the 5-tensor input of the next FlowNetS
    (flownet_cs.py:21-36, flownet_css.py:21-36)."""
    a32 = np.asarray(inputs["input_a"], np.float32)
    b32 = np.asarray(inputs["input_b"], np.float32)
    warped = ops.flow_warp(b32, flow.astype(np.float32)).astype(F64)
    berr = nn.channel_norm(a32.astype(F64) - warped)
    return {"input_a": inputs["input_a"], "input_b": inputs["input_b"], "warped": warped,
            "flow": flow * 0.05, "brightness_error": berr}


def flownet_cs(weights, inputs, scope="FlowNetCS"):
    """FlowNetCS.model (flownet_cs.py:15-38)."""
    pc = flownet_c(weights, inputs, scope + "/FlowNetC")
    return flownet_s(weights, _stack_inputs(inputs, pc["flow"]), scope + "/FlowNetS")


def flownet_css(weights, inputs, scope="FlowNetCSS"):
    """FlowNetCSS.model (flownet_css.py:15-38)."""
    pcs = flownet_cs(weights, inputs, scope + "/FlowNetCS")
    return flownet_s(weights, _stack_inputs(inputs, pcs["flow"]), scope + "/FlowNetS")


def flownet2(weights, inputs, scope="FlowNet2"):
    """FlowNet2.model (flownet2.py:18-105)."""
    a = np.asarray(inputs["input_a"], F64)
    b32 = np.asarray(inputs["input_b"], np.float32)
    _, H, W, _ = a.shape
    css = flownet_css(weights, inputs, scope + "/FlowNetCSS")["flow"]
    sd = flownet_sd(weights, inputs, scope + "/FlowNetSD")["flow"]
    diff_sd = nn.channel_norm(a - ops.flow_warp(b32, sd.astype(np.float32)).astype(F64))  # :33-35
    diff_css = nn.channel_norm(a - ops.flow_warp(b32, css.astype(np.float32)).astype(F64))  # :37-39
    x = np.concatenate([a, sd, css, nn.channel_norm(sd), nn.channel_norm(css),
                        diff_sd, diff_css], axis=3)  # 11 channels, :41-47
    sc = _Scope(weights, scope)
    f0 = sc.conv(x, "fuse_conv0")
    f1_1 = sc.conv(sc.conv(f0, "fuse_conv1", 2), "fuse_conv1_1")
    f2_1 = sc.conv(sc.conv(f1_1, "fuse_conv2", 2), "fuse_conv2_1")
    pf2 = sc.conv(f2_1, "predict_flow2", act=False)
    # the fusion arg_scope (:50-57) sets no biases_initializer=None: all four transposed convs have biases
    cat1 = np.concatenate([f1_1, sc.deconv(f2_1, "fuse_deconv1", bias=True),
                           sc.deconv(pf2, "fuse_upsample_flow2to1", act=False, bias=True)], axis=3)
    pf1 = sc.conv(sc.conv(cat1, "fuse_interconv1", act=False), "predict_flow1", act=False)
    cat0 = np.concatenate([f0, sc.deconv(cat1, "fuse_deconv0", bias=True),
                           sc.deconv(pf1, "fuse_upsample_flow1to0", act=False, bias=True)], axis=3)
    pf0 = sc.conv(sc.conv(cat0, "fuse_interconv0", act=False), "predict_flow0", act=False)
    flow = nn.resize_bilinear_align_corners(pf0, (H, W))  # identity size, :100-101
    return {"predict_flow0": pf0, "flow": flow}


MODELS = {"FlowNetS": flownet_s, "FlowNetC": flownet_c, "FlowNetSD": flownet_sd,
          "FlowNetCS": flownet_cs, "FlowNetCSS": flownet_css, "FlowNet2": flownet2,
          "FlowNetS_interp": flownet_s_interp}


# ----------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------
def average_endpoint_error(labels, predictions):
    """utils.py:209-224 -- sum over pixels of the per-pixel L2 norm, divided by
    the batch size only."""
    d = np.asarray(predictions, F64) - np.asarray(labels, F64)
    return float(np.sqrt(np.sum(d * d, axis=3)).sum() / d.shape[0])


def average_endpoint_error_hfem(labels, predictions, add_hfem="", lambda_w=2.0, perc_hfem=50, edges=None):
    """utils.py:227-339.  '' -> plain AEPE (sum of per-pixel EPE / batch).  'hard': the top round(perc/100 * #pixels)
    EPE values (tf.nn.top_k over the whole flattened batch, :268-275) summed with weight (1 + lambda), divided by
    the batch size and multiplied by #pixels / #hard pixels (:305-312).  'edges' (with an edge map in [0,1], same
    N x H x W x 1 shape): sum of epe * (1 + lambda * edges) / batch (:315-324).  Anything else: plain AEPE."""
    d = np.asarray(labels, F64) - np.asarray(predictions, F64)
    epe = np.sqrt(np.sum(d * d, axis=3, keepdims=True))
    n = d.shape[0]
    mode = (add_hfem or "").lower()
    if mode == "hard":
        flat = epe.reshape(-1)
        k = int(np.round(np.float32(perc_hfem / 100) * np.float32(flat.size)))  # tf.round: half to even
        hard = np.sort(flat)[::-1][:k]
        return float((1.0 + lambda_w) * hard.sum() / n * (flat.size / max(k, 1)))
    if mode == "edges" and edges is not None:
        return float((epe + lambda_w * epe * np.asarray(edges, F64)).sum() / n)
    return float(epe.sum() / n)


def mean_endpoint_error(gt_flow, pred_flow):
    """utils.py:342-355: mean over all pixels and samples of the per-pixel L2 norm."""
    d = np.asarray(gt_flow, F64) - np.asarray(pred_flow, F64)
    return float(np.sqrt(np.sum(d * d, axis=3)).mean())


def multiscale_loss(gt_flow, preds, weights=None, scope="FlowNetS", l2=4e-4, gt_scale=0.05):
    """FlowNetS.loss (flownet_s.py:122-161).  compute_weighted_loss with TF's
    default SUM_BY_NONZERO_WEIGHTS over the 5 scalars => (sum w_i L_i)/5 (UNPINNED,
    SURVEY.md A.5), plus the slim L2 regularisers (0.5*l2*|W|^2 over slim.conv2d
    weights of ``scope``; conv-transpose weights and biases are not regularised)."""
    flow = np.asarray(gt_flow, np.float32) * np.float32(gt_scale)
    losses = []
    for lvl in (6, 5, 4, 3, 2):
        p = preds[f"predict_flow{lvl}"]
        losses.append(average_endpoint_error(ops.downsample(flow, p.shape[1:3]), p))
    data = sum(w * l for w, l in zip(LOSS_WEIGHTS, losses)) / 5.0
    reg = 0.0
    if weights is not None:
        for name, w in weights.items():
            if name.startswith(scope + "/") and name.endswith("/weights") \
                    and "deconv" not in name and "upsample_flow" not in name:
                reg += 0.5 * l2 * float(np.sum(np.square(np.asarray(w, F64))))
    return data + reg, losses
