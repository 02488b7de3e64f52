"""Training losses of the reference (FlowNetS.loss, src/flownet_s/flownet_s.py:122-161;
average_endpoint_error, src/utils.py:209-224; FlowNet2.loss, src/flownet2/flownet2.py:107-116).
The NaN-aware ground-truth downsample is the HIP op; the scalar reductions are torch device ops."""
import numpy as np
import torch

from .downsample import downsample

LOSS_WEIGHTS = (0.32, 0.08, 0.02, 0.01, 0.005)


def average_endpoint_error(labels, predictions):
    """sum over pixels of ||pred - label||_2, divided by the batch size only."""
    n = predictions.shape[0]
    d = predictions.float() - labels.float()
    return torch.sqrt((d * d).sum(dim=3)).sum() / n


def _as_dev(x, like):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x), device=like.device)


def multiscale_loss(flow, predictions, weights=None, scope=None, l2=4e-4, gt_scale=0.05):
    p6 = predictions['predict_flow6']
    flow = _as_dev(flow, p6).to(device=p6.device, dtype=torch.float32) * gt_scale
    losses = []
    for lvl in (6, 5, 4, 3, 2):
        p = predictions['predict_flow%d' % lvl]
        losses.append(average_endpoint_error(downsample(flow, [p.shape[1], p.shape[2]]), p))
    # tf.losses.compute_weighted_loss(list, weights) with SUM_BY_NONZERO_WEIGHTS: (sum w_i L_i) / 5
    total = sum(w * l for w, l in zip(LOSS_WEIGHTS, losses)) / 5.0
    if weights is not None and scope is not None:
        reg = 0.0
        for name, w in weights.items():
            if name.startswith(scope + "/") and name.endswith("/weights") \
                    and "deconv" not in name and "upsample_flow" not in name:
                reg += 0.5 * l2 * float(np.sum(np.square(np.asarray(w, np.float64))))
        total = total + reg
    return total


def fusion_loss(flow, predictions):
    p0 = predictions['predict_flow0']
    flow = _as_dev(flow, p0).to(device=p0.device, dtype=torch.float32)
    return average_endpoint_error(downsample(flow, [p0.shape[1], p0.shape[2]]), p0)
