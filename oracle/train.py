"""Oracle (test infrastructure): FlowNetS / FlowNetSD / FlowNetS_interp loss gradients and the Adam update on CPU.

The gradient of the reference's loss graph (what tf.gradients produces) is obtained by running the same
graph in torch float64 on the CPU and using autograd; the graphs are the ones restated in oracle/models.py
(flownet_s.py:14-161, flownet_sd.py:14-160; FlowNetS_interp = the S tower on [image | 0.05 sparse flow | matches]
without head biases, flownet_s_interp.py:21-157, with the hard-flow-example-mining losses of utils.py:227-339) and
tests check this torch forward against that NumPy forward on the same weights.
PARITY UNPINNED like oracle/nn.py (TensorFlow is not available).  The Adam update restates
tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v moments; w -= lr_t*m/(sqrt(v)+eps).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops

LOSS_WEIGHTS = {6: 0.32, 5: 0.08, 4: 0.02, 3: 0.01, 2: 0.005}


def _leaky(x, positive=None):
    """utils.py:401-405: 0.55 x + 0.45 |x|.  ``positive``: the branch the device took per element -- a bool array
    (True = output > 0) or a sign array (+1 / -1 / 0: an output that is EXACTLY 0 differentiates as 0.55, tf.abs' =
    sign(0) = 0, which is also what the device's backward pass applies)."""
    if positive is not None:
        sg = torch.as_tensor(np.asarray(positive)).permute(0, 3, 1, 2)
        if sg.dtype == torch.bool:
            return torch.where(sg, x, 0.1 * x)
        return torch.where(sg > 0, x, torch.where(sg < 0, 0.1 * x, 0.55 * x))
    return 0.55 * x + 0.45 * x.abs()  # utils.py:401-405


def flownet_s_loss_and_grads(weights, input_a, input_b, gt_flow, scope="FlowNetS", l2=0.0, act_grads=None, signs=None,
                             model="FlowNetS", add_hfem="", lambda_w=2.0, perc_hfem=50, edges=None, stacked=None):
    """Returns (loss, {variable name: gradient in the reference layout}, predictions).  l2 > 0 adds the slim
    regulariser 0.5*l2*|W|^2 of the slim.conv2d weights to the loss (and so l2*W to their gradients).
    act_grads: optional dict, filled with {layer name: dLoss/d(layer output), NHWC} for debugging.
    signs: optional {layer name: bool NHWC array, True where the layer output is positive}.  LeakyReLU has a
    kink at 0; a pre-activation of magnitude ~1e-7 lands on either side depending on fp32 rounding, and the
    two branches have gradients 1 and 0.1.  Passing the branch the device actually took makes both sides
    differentiate the same piecewise-linear function (tests bound how many such elements there are)."""
    P = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True) for k, v in weights.items()
         if k.startswith(scope + "/")}

    def conv(x, name, stride=1, pad=1, act=True):
        w = P[f"{scope}/{name}/weights"].permute(3, 2, 0, 1)
        y = F.conv2d(x, w, P.get(f"{scope}/{name}/biases"), stride=stride, padding=pad)  # interp heads: no bias
        if act_grads is not None:
            acts[name + "/pre"] = y
        y = _leaky(y, None if signs is None else signs.get(name)) if act else y
        if act_grads is not None:
            y.retain_grad()
            acts[name] = y
        return y

    def deconv(x, name, act=True):
        w = P[f"{scope}/{name}/weights"].permute(3, 2, 0, 1)  # [ky,kx,o,i] -> [i,o,ky,kx]
        # biases_initializer=None in S / SD (flownet_s.py:53, flownet_sd.py:44); FlowNetS_interp built with
        # no_deconv_biases=False gives deconvN -- never upsample_flowXtoY -- a bias (flownet_s_interp.py:84-126)
        b = P.get(f"{scope}/{name}/biases") if (model == "FlowNetS_interp" and name.startswith("deconv")) else None
        y = F.conv_transpose2d(x, w, b, stride=2, padding=1)
        if act_grads is not None:
            acts[name + "/pre"] = y
        return _leaky(y, None if signs is None else signs.get(name)) if act else y

    acts = {}
    a = torch.tensor(np.asarray(input_a, np.float64)).permute(0, 3, 1, 2)
    b = torch.tensor(np.asarray(input_b, np.float64)).permute(0, 3, 1, 2)
    x = torch.cat([a, b], 1)
    if stacked is not None:
        # the last FlowNetS of FlowNetCS / CSS: its 12-channel input [a | b | warped | 0.05 flow | brightness error]
        # (flownet_cs.py:21-36) comes from networks built trainable=False -- a constant of the differentiation
        x = torch.tensor(np.asarray(stacked, np.float64)).permute(0, 3, 1, 2)
    sd = model == "FlowNetSD"  # flownet_sd.py:14-119: all-3x3 encoder with conv0 at full resolution, interconvN heads
    if sd:
        c0 = conv(x, "conv0")
        c1_1 = conv(conv(c0, "conv1", 2), "conv1_1")
        c2 = conv(c1_1, "conv2", 2)                                   # the level-2 skip is conv2 (flownet_sd.py:97)
        c3_1 = conv(conv(conv(c2, "conv2_1"), "conv3", 2), "conv3_1")
    else:
        c1 = conv(x, "conv1", 2, 3)
        c2 = conv(c1, "conv2", 2, 2)
        c3_1 = conv(conv(c2, "conv3", 2, 2), "conv3_1")
    c4_1 = conv(conv(c3_1, "conv4", 2), "conv4_1")
    c5_1 = conv(conv(c4_1, "conv5", 2), "conv5_1")
    c6_1 = conv(conv(c5_1, "conv6", 2), "conv6_1")
    preds = {6: conv(c6_1, "predict_flow6", act=False)}
    cur = c6_1
    for lvl, skip in zip((5, 4, 3, 2), (c5_1, c4_1, c3_1, c2)):
        cur = torch.cat([skip, deconv(cur, f"deconv{lvl}"),
                         deconv(preds[lvl + 1], f"upsample_flow{lvl + 1}to{lvl}", act=False)], 1)
        head_in = conv(cur, f"interconv{lvl}", act=False) if sd else cur
        preds[lvl] = conv(head_in, f"predict_flow{lvl}", act=False)
    n = a.shape[0]
    # labels: FlowNetS 0.05 * gt (flownet_s.py:123), FlowNetSD 20 * gt (flownet_sd.py:122)
    gt = np.asarray(gt_flow, np.float32) * np.float32(20.0 if sd else 0.05)
    loss = 0.0
    for lvl, wgt in LOSS_WEIGHTS.items():
        p = preds[lvl]
        label = torch.tensor(ops.downsample(gt, (p.shape[2], p.shape[3])).astype(np.float64)).permute(0, 3, 1, 2)
        epe = torch.sqrt(((p - label) ** 2).sum(1))
        mode = (add_hfem or "").lower()
        if mode == "hard":  # utils.py:227-312: the top-k EPE values of the batch, (1 + lambda), rescaled by #pixels / k
            flat = epe.reshape(-1)
            k = int(np.round(np.float32(perc_hfem / 100) * np.float32(flat.numel())))
            loss = loss + wgt * (1.0 + lambda_w) * torch.topk(flat, k).values.sum() / n * (flat.numel() / max(k, 1))
        elif mode == "edges" and edges is not None:  # utils.py:313-320, edges downsampled per scale (flownet_s_interp.py)
            e = torch.tensor(ops.downsample(np.asarray(edges, np.float32), (p.shape[2], p.shape[3])).astype(np.float64))[..., 0]
            loss = loss + wgt * (epe + lambda_w * epe * e).sum() / n
        else:
            loss = loss + wgt * epe.sum() / n
    loss = loss / 5.0  # compute_weighted_loss, SUM_BY_NONZERO_WEIGHTS over the five scalars
    if l2 > 0:
        for k, v in P.items():
            if k.endswith("/weights") and "deconv" not in k and "upsample_flow" not in k:
                loss = loss + 0.5 * l2 * (v ** 2).sum()
    loss.backward()
    grads = {k: v.grad.numpy() for k, v in P.items() if v.grad is not None}
    if act_grads is not None:
        act_grads.update({k: v.grad.permute(0, 2, 3, 1).numpy() for k, v in acts.items() if not k.endswith("/pre")})
        act_grads.update({k + "/value": v.detach().permute(0, 2, 3, 1).numpy() for k, v in acts.items()})
    out = {f"predict_flow{l}": p.detach().permute(0, 2, 3, 1).numpy() for l, p in preds.items()}
    return float(loss.detach()), grads, out


def correlation_torch(a, b, md=20, s2=2):
    """The FlowNetC call of the correlation op (kernel 1, stride_1 1, pad = max_displacement; correlation_kernel.cu.cc:
    45-110) on NCHW float64 torch tensors, differentiable: out[:, (p + r) * g + (o + r)] = mean_c a * shift(b, s2 p, s2 o)."""
    r = md // s2
    g = 2 * r + 1
    n, c, h, w = a.shape
    bp = F.pad(b, (md, md, md, md))
    outs = []
    for p in range(-r, r + 1):
        for o in range(-r, r + 1):
            y0, x0 = md + s2 * p, md + s2 * o
            outs.append((a * bp[:, :, y0:y0 + h, x0:x0 + w]).sum(1) / c)
    return torch.stack(outs, 1)


def flownet_c_loss_and_grads(weights, input_a, input_b, gt_flow, scope="FlowNetC", signs=None):
    """FlowNetC.model + FlowNetC.loss (flownet_c.py:15-170) in torch float64 with autograd: shared towers (one set of
    conv1-3 variables, reuse=True at :34-37, so their gradients sum over both towers), correlation + LeakyReLU,
    conv_redir, the FlowNetS-style rest and refinement; labels 0.05 * gt, five scales, / 5.  signs: {layer name (tower
    b: name + '_b'; 'correlation'): sign of the layer output as the device computed it}, see flownet_s_loss_and_grads."""
    P = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True) for k, v in weights.items()
         if k.startswith(scope + "/")}

    def sg(name):
        return None if signs is None else signs.get(name)

    def conv(x, name, stride=1, pad=1, act=True, tag=None):
        y = F.conv2d(x, P[f"{scope}/{name}/weights"].permute(3, 2, 0, 1), P.get(f"{scope}/{name}/biases"), stride=stride, padding=pad)
        return _leaky(y, sg(tag or name)) if act else y

    def deconv(x, name, act=True):
        y = F.conv_transpose2d(x, P[f"{scope}/{name}/weights"].permute(3, 2, 0, 1), stride=2, padding=1)
        return _leaky(y, sg(name)) if act else y

    a = torch.tensor(np.asarray(input_a, np.float64)).permute(0, 3, 1, 2)
    b = torch.tensor(np.asarray(input_b, np.float64)).permute(0, 3, 1, 2)
    a2 = conv(conv(a, "conv1", 2, 3), "conv2", 2, 2)
    a3 = conv(a2, "conv3", 2, 2)
    b3 = conv(conv(conv(b, "conv1", 2, 3, tag="conv1_b"), "conv2", 2, 2, tag="conv2_b"), "conv3", 2, 2, tag="conv3_b")
    cc = _leaky(correlation_torch(a3, b3), sg("correlation"))
    net = torch.cat([conv(a3, "conv_redir", 1, 0), cc], 1)
    c3_1 = conv(net, "conv3_1")
    c4_1 = conv(conv(c3_1, "conv4", 2), "conv4_1")
    c5_1 = conv(conv(c4_1, "conv5", 2), "conv5_1")
    c6_1 = conv(conv(c5_1, "conv6", 2), "conv6_1")
    preds = {6: conv(c6_1, "predict_flow6", act=False)}
    cur = c6_1
    for lvl, skip in zip((5, 4, 3, 2), (c5_1, c4_1, c3_1, a2)):
        cur = torch.cat([skip, deconv(cur, f"deconv{lvl}"),
                         deconv(preds[lvl + 1], f"upsample_flow{lvl + 1}to{lvl}", act=False)], 1)
        preds[lvl] = conv(cur, f"predict_flow{lvl}", act=False)
    n = a.shape[0]
    gt = np.asarray(gt_flow, np.float32) * np.float32(0.05)
    loss = 0.0
    for lvl, wgt in LOSS_WEIGHTS.items():
        p = preds[lvl]
        label = torch.tensor(ops.downsample(gt, (p.shape[2], p.shape[3])).astype(np.float64)).permute(0, 3, 1, 2)
        loss = loss + wgt * torch.sqrt(((p - label) ** 2).sum(1)).sum() / n
    loss = loss / 5.0
    loss.backward()
    grads = {k: v.grad.numpy() for k, v in P.items() if v.grad is not None}
    return float(loss.detach()), grads, {f"predict_flow{l}": p.detach().permute(0, 2, 3, 1).numpy() for l, p in preds.items()}


def fusion_loss_and_grads(weights, fusion_input, gt_flow, scope="FlowNet2", signs=None):
    """The trainable part of FlowNet2 (flownet2.py:50-116): the fusion network on its 11-channel input (a constant:
    CSS and SD are built trainable=False, :22-23) and FlowNet2.loss = average_endpoint_error(downsample(flow, size of
    predict_flow0), predict_flow0) -- unscaled ground truth, weight 1.  The four transposed convs carry biases.
    Returns (loss, {variable: gradient in the reference layout}, predict_flow0 NHWC)."""
    P = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True) for k, v in weights.items()
         if k.startswith(scope + "/fuse_") or k.startswith(scope + "/predict_flow")}

    def conv(x, name, stride=1, act=True):
        y = F.conv2d(x, P[f"{scope}/{name}/weights"].permute(3, 2, 0, 1), P[f"{scope}/{name}/biases"], stride=stride, padding=1)
        return _leaky(y, None if signs is None else signs.get(name)) if act else y

    def deconv(x, name, act=True):
        y = F.conv_transpose2d(x, P[f"{scope}/{name}/weights"].permute(3, 2, 0, 1), P[f"{scope}/{name}/biases"],
                               stride=2, padding=1)
        return _leaky(y, None if signs is None else signs.get(name)) if act else y

    x = torch.tensor(np.asarray(fusion_input, np.float64)).permute(0, 3, 1, 2)
    f0 = conv(x, "fuse_conv0")
    f1_1 = conv(conv(f0, "fuse_conv1", 2), "fuse_conv1_1")
    f2_1 = conv(conv(f1_1, "fuse_conv2", 2), "fuse_conv2_1")
    pf2 = conv(f2_1, "predict_flow2", act=False)
    cat1 = torch.cat([f1_1, deconv(f2_1, "fuse_deconv1"), deconv(pf2, "fuse_upsample_flow2to1", act=False)], 1)
    pf1 = conv(conv(cat1, "fuse_interconv1", act=False), "predict_flow1", act=False)
    cat0 = torch.cat([f0, deconv(cat1, "fuse_deconv0"), deconv(pf1, "fuse_upsample_flow1to0", act=False)], 1)
    pf0 = conv(conv(cat0, "fuse_interconv0", act=False), "predict_flow0", act=False)
    gt = np.asarray(gt_flow, np.float32)
    label = torch.tensor(ops.downsample(gt, (pf0.shape[2], pf0.shape[3])).astype(np.float64)).permute(0, 3, 1, 2)
    loss = torch.sqrt(((pf0 - label) ** 2).sum(1)).sum() / x.shape[0]
    loss.backward()
    grads = {k: v.grad.numpy() for k, v in P.items() if v.grad is not None}
    return float(loss.detach()), grads, pf0.detach().permute(0, 2, 3, 1).numpy()


def adam_update(w, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    lr_t = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    return w - lr_t * m / (np.sqrt(v) + eps), m, v
