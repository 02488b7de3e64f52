"""Training input pipeline of the reference (src/dataloader.py:383-560 `load_batch`, src/dataset_configs.py)
re-expressed for list files: the reference reads TFRecords it cannot ship; here a text file lists
`image_a image_b flow.flo` per line (the same triples the TFRecords were built from), samples are batched on the
host, and the augmentation runs on the GPU through the preprocessing plugin's op surface (src/preprocessing.py):
DataAugmentation on both images (image b inherits image a's transform), FlowAugmentation on the ground truth.

PREPROCESS values are the reference's FlyingChairs configuration (dataset_configs.py:60-157) as data.
"""
import numpy as np
import torch

from . import preprocessing as P
from .flowlib import read_flow


def _p(rand_type, exp, mean, spread, prob=1.0):
    return {"rand_type": rand_type, "exp": exp, "mean": mean, "spread": spread, "prob": prob}


FLYING_CHAIRS_PREPROCESS = {
    "scale": False,
    "crop_height": 384,
    "crop_width": 448,
    "image_a": {
        "translate": _p("uniform_bernoulli", False, 0, 0.4),
        "rotate": _p("uniform_bernoulli", False, 0, 0.4),
        "zoom": _p("uniform_bernoulli", True, 0.2, 0.4),
        "squeeze": _p("uniform_bernoulli", True, 0, 0.3),
        "noise": _p("uniform_bernoulli", False, 0.03, 0.03),
    },
    # all preprocessing of image a is applied to image b in addition to the following
    "image_b": {
        "translate": _p("gaussian_bernoulli", False, 0, 0.03),
        "rotate": _p("gaussian_bernoulli", False, 0, 0.03),
        "zoom": _p("gaussian_bernoulli", True, 0, 0.03),
        "gamma": _p("gaussian_bernoulli", True, 0, 0.02),
        "brightness": _p("gaussian_bernoulli", False, 0, 0.02),
        "contrast": _p("gaussian_bernoulli", True, 0, 0.02),
        "color": _p("gaussian_bernoulli", True, 0, 0.02),
        "coeff_schedule_param": {"half_life": 50000, "initial_coeff": 0.5, "final_coeff": 1},
    },
}


def config_to_arrays(dataset_config):
    """dataloader.py:279-308: the per-parameter dict -> the parallel attribute lists of the op."""
    out = {"name": [], "rand_type": [], "exp": [], "mean": [], "spread": [], "prob": [], "coeff_schedule": []}
    for name, value in dataset_config.items():
        if name == "coeff_schedule_param":
            out["coeff_schedule"] = [value["half_life"], value["initial_coeff"], value["final_coeff"]]
        else:
            out["name"].append(name)
            for k in ("rand_type", "exp", "mean", "spread", "prob"):
                out[k].append(value[k])
    return out


def read_list(path):
    with open(path) as f:
        rows = [ln.split() for ln in f.read().splitlines() if ln.strip()]
    for r in rows:
        if len(r) != 3:
            raise ValueError("expected `image_a image_b flow.flo` per line, got: %r" % (r,))
    return rows


def load_batches(list_path, batch_size, preprocess=FLYING_CHAIRS_PREPROCESS, data_augmentation=True, seed=0,
                 global_step=0, epochs=None):
    """Generator of (image_a, image_b, flow) device tensors [B,h,w,3|3|2]; images in [0,1].  With
    data_augmentation the crop is (crop_height, crop_width) and the flow is transformed with the two augmentation
    matrices (dataloader.py:480-540); without it the samples pass through unchanged."""
    from .net import imread
    rows = read_list(list_path)
    rng = np.random.default_rng(seed)
    a_cfg, b_cfg = config_to_arrays(preprocess["image_a"]), config_to_arrays(preprocess["image_b"])
    crop = (preprocess["crop_height"], preprocess["crop_width"])
    epoch, step = 0, int(global_step)
    while epochs is None or epoch < epochs:
        order = rng.permutation(len(rows))
        for i in range(0, len(order) - batch_size + 1, batch_size):
            pick = [rows[j] for j in order[i:i + batch_size]]
            a = np.stack([imread(r[0]) for r in pick]).astype(np.float32) / 255.0
            b = np.stack([imread(r[1]) for r in pick]).astype(np.float32) / 255.0
            f = np.stack([read_flow(r[2]) for r in pick])
            if not data_augmentation:
                dev = lambda x: torch.from_numpy(x).to(P._hip.require_device())
                yield dev(a), dev(b), dev(f)
            else:
                oa, ob, ta, itb = P.data_augmentation(
                    a, b, step, crop, a_cfg["name"], a_cfg["rand_type"], a_cfg["exp"], a_cfg["mean"], a_cfg["spread"],
                    a_cfg["prob"], a_cfg["coeff_schedule"], b_cfg["name"], b_cfg["rand_type"], b_cfg["exp"], b_cfg["mean"],
                    b_cfg["spread"], b_cfg["prob"], b_cfg["coeff_schedule"], seed=int(rng.integers(1 << 31)))
                if "noise" in preprocess["image_a"]:
                    # the plugin leaves noise to the Python side (augmentation_base.h:126): gaussian, sigma drawn per batch
                    n = preprocess["image_a"]["noise"]
                    sigma = abs(P.rng_generate(rng, n, 1.0, 0.0))
                    g = torch.Generator(device=oa.device).manual_seed(int(rng.integers(1 << 31)))
                    oa = (oa + sigma * torch.randn(oa.shape, generator=g, device=oa.device)).clamp_(0.0, 1.0)
                    ob = (ob + sigma * torch.randn(ob.shape, generator=g, device=ob.device)).clamp_(0.0, 1.0)
                yield oa, ob, P.flow_augmentation(f, ta, itb, crop)
            step += 1
        epoch += 1
