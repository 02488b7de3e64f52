"""Harness: the parts of /root/reference src/net.py that sit on the inference path
(``Net.test`` :484-628, ``adapt_x`` :324-392, ``postproc_y_hat_test`` :463-473,
``get_padded_image_size`` :301-309), re-expressed over the HIP engine.

Deviations from the reference (its classic entry points are stale, SURVEY.md
Appendix C): ``test`` works for all six classic models (defect D1), the output
sub-folder is the parent directory name of ``input_a`` (the intended behaviour
behind defect D3), and a missing checkpoint falls back -- loudly -- to seeded
synthetic weights, because no trained weights can be obtained offline.
"""
import os
import sys
from enum import Enum
from math import ceil

import numpy as np
import torch

from . import weights as W
from .engine import Engine
from .flowlib import flow_to_image, read_flow, write_flow
from .training_schedules import LONG_SCHEDULE


class Mode(Enum):
    TRAIN = 1
    TEST = 2


def imread(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def imsave(path, img):
    from PIL import Image
    Image.fromarray(np.asarray(img, np.uint8)).save(path)


class Net(object):
    model_name = None  # set by subclasses: 'FlowNetS', ...

    def __init__(self, mode=Mode.TRAIN, debug=False, dtype="f32"):
        self.mode = mode
        self.debug = debug
        self.dtype = dtype
        self.weights = None
        self._engines = {}

    # ---- weights -----------------------------------------------------------------------------
    def load_weights(self, checkpoint=None, seed=1234):
        """``checkpoint``: an .npz keyed by the reference's variable names (weights.load_npz).
        A TF checkpoint prefix that does not exist here -> seeded synthetic weights + warning."""
        if checkpoint is not None and os.path.exists(checkpoint) and checkpoint.endswith(".npz"):
            self.weights = W.load_npz(checkpoint)
        else:
            if checkpoint is not None:
                sys.stderr.write("WARNING: checkpoint %r not found or not .npz; using seeded synthetic "
                                 "weights (seed %d) -- flows are NOT meaningful predictions\n" % (checkpoint, seed))
            self.weights = W.init_weights(self.model_name, seed)
        self._engines = {}
        return self.weights

    def engine(self, batch, height, width):
        if self.weights is None:
            self.load_weights()
        key = (batch, height, width, self.dtype)
        if key not in self._engines:
            self._engines[key] = Engine(self.model_name, self.weights, batch, height, width, self.dtype)
        return self._engines[key]

    # ---- graph: same signature as the reference's model()/loss() --------------------------------
    def model(self, inputs, training_schedule=LONG_SCHEDULE, trainable=True):
        """inputs: {'input_a','input_b'} NHWC float32 in [0,1], H and W multiples of 64.
        Returns the reference's prediction dict (predict_flow6..2 / predict_flow0, 'flow') as
        fp32 ROCm tensors."""
        a, b = inputs['input_a'], inputs['input_b']
        n, h, w, _ = a.shape
        eng = self.engine(int(n), int(h), int(w))
        out = eng(a, b)
        return {k: v.clone() for k, v in out.items()}

    # ---- test-time input adaptation ---------------------------------------------------------------
    def get_padded_image_size(self, og_height, og_width, divisor=64):
        return int(ceil(og_height / divisor) * divisor), int(ceil(og_width / divisor) * divisor)

    def adapt_x(self, input_a, input_b, divisor=64):
        """[0,255] -> [0,1] when max > 1; add the batch axis; zero-pad bottom/right to a multiple
        of ``divisor``.  Returns (a, b, original_shape_or_None)."""
        def norm(x):
            x = np.asarray(x)
            return x / 255.0 if x.max() > 1.0 else x.astype(np.float64)

        a, b = norm(input_a), norm(input_b)
        if a.shape != b.shape:
            raise AssertionError("FATAL: image dimensions do not match. Image 1 has shape: {0}, "
                                 "Image 2 has shape: {1}".format(a.shape, b.shape))
        if a.ndim == 3:
            a, b = a[None], b[None]
        h, w = a.shape[1:3]
        nh, nw = self.get_padded_image_size(h, w, divisor)
        info = None
        if (nh, nw) != (h, w):
            info = a.shape
            pad = [(0, 0), (0, nh - h), (0, nw - w), (0, 0)]
            a, b = np.pad(a, pad), np.pad(b, pad)
        return a.astype(np.float32), b.astype(np.float32), info

    def postproc_y_hat_test(self, pred_flows, adapt_info=None):
        if adapt_info is not None:
            pred_flows = pred_flows[0:adapt_info[-3], 0:adapt_info[-2], :]
        return pred_flows

    # ---- single-pair inference ----------------------------------------------------------------------
    def test(self, checkpoint, input_a_path, input_b_path=None, out_path='./', save_image=True, save_flo=True,
             compute_metrics=True, gt_flow=None, new_par_folder=None):
        a, b, info = self.adapt_x(imread(input_a_path), imread(input_b_path))
        if self.weights is None:
            self.load_weights(checkpoint)
        preds = self.model({'input_a': a, 'input_b': b}, LONG_SCHEDULE, trainable=False)
        flow = preds['flow'][0].float().cpu().numpy()
        y_info = (info[-3], info[-2], 2) if info is not None else None
        flow = self.postproc_y_hat_test(flow, y_info)

        parent = new_par_folder if new_par_folder is not None else \
            os.path.basename(os.path.dirname(os.path.abspath(input_a_path)))
        unique_name = os.path.splitext(os.path.basename(input_a_path))[0]
        out_dir = os.path.join(out_path, parent)
        max_flow = -1
        gt = None
        if compute_metrics and gt_flow is not None:
            gt = read_flow(gt_flow)
            max_flow = float(np.max(np.sqrt(gt[:, :, 0] ** 2 + gt[:, :, 1] ** 2)))
        if save_image or save_flo:
            os.makedirs(out_dir, exist_ok=True)
        if save_image:
            full = os.path.join(out_dir, unique_name + '_viz.png')
            imsave(full, flow_to_image(flow.copy()))
            imsave(full.replace('.png', '_norm_gt_max_motion.png'), flow_to_image(flow.copy(), maxflow=max_flow))
        if save_flo:
            write_flow(flow, os.path.join(out_dir, unique_name + '_flow.flo'))
        if gt is not None:
            from .flowlib import endpoint_error
            print("{}: EPE all = {:.4f}".format(unique_name, endpoint_error(flow, gt)))
        return flow
