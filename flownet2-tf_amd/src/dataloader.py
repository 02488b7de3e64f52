"""Training input pipeline of the reference (src/dataloader.py:383-560 `load_batch`, src/dataset_configs.py)
over two sample sources: the reference's ZLIB TFRecord files (src/tfrecord.py reads them without TensorFlow) or a
text file listing `image_a image_b flow.flo` per line (the triples the TFRecords were built from); samples are
batched on the host, and the augmentation runs on the GPU through the preprocessing plugin's op surface (src/preprocessing.py):
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


def _list_epoch(rows, rng):
    """One epoch of (image_a, image_b, flow) host arrays from a list file, in a fresh random order."""
    from .net import imread
    for j in rng.permutation(len(rows)):
        r = rows[j]
        yield imread(r[0]).astype(np.float32) / 255.0, imread(r[1]).astype(np.float32) / 255.0, read_flow(r[2])


def _tfrecord_epoch(path, rng, image_size, scale, shuffle_buffer):
    """One epoch over a TFRecord file of the reference's samples (src/tfrecord.py), shuffled through a bounded
    buffer like the reference's slim DatasetDataProvider queue (dataloader.py:444-451: common_queue_capacity)."""
    from . import tfrecord
    buf = []
    div = 255.0 if scale else 1.0  # records hold images already in [0,1] unless PREPROCESS['scale'] (dataloader.py:466)
    for smp in tfrecord.read_samples(path, image_size[0], image_size[1]):
        item = (smp["image_a"] / div, smp["image_b"] / div, smp["flow"])
        if len(buf) < shuffle_buffer:
            buf.append(item)
            continue
        k = int(rng.integers(len(buf)))
        out, buf[k] = buf[k], item
        yield out
    for k in rng.permutation(len(buf)):
        yield buf[k]


def is_tfrecord(path):
    return str(path).endswith((".tfrecords", ".tfrecord"))


def load_batches(list_path, batch_size, preprocess=FLYING_CHAIRS_PREPROCESS, data_augmentation=True, seed=0,
                 global_step=0, epochs=None, image_size=(384, 512), shuffle_buffer=256):
    """Generator of (image_a, image_b, flow) device tensors [B,h,w,3|3|2]; images in [0,1].  ``list_path`` is a
    text file of `image_a image_b flow.flo` triples or a ``.tfrecords`` file as the reference's converter writes it
    (``image_size`` = the dataset's PADDED_IMAGE_HEIGHT/WIDTH, dataset_configs.py:42-43).  With data_augmentation
    the crop is (crop_height, crop_width) and the flow is transformed with the two augmentation matrices
    (dataloader.py:480-540); without it the samples pass through unchanged."""
    rng = np.random.default_rng(seed)
    rows = None if is_tfrecord(list_path) else read_list(list_path)
    a_cfg, b_cfg = config_to_arrays(preprocess["image_a"]), config_to_arrays(preprocess["image_b"])
    crop = (preprocess["crop_height"], preprocess["crop_width"])
    epoch, step = 0, int(global_step)
    while epochs is None or epoch < epochs:
        source = _list_epoch(rows, rng) if rows is not None else _tfrecord_epoch(
            list_path, rng, image_size, bool(preprocess.get("scale", False)), shuffle_buffer)
        pick = []
        produced = False
        for smp in source:
            pick.append(smp)
            if len(pick) < batch_size:
                continue
            a, b, f = (np.stack([p[i] for p in pick]).astype(np.float32) for i in range(3))
            pick = []
            produced = True
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
        if not produced:
            raise ValueError("%s holds fewer than one batch (%d) of samples" % (list_path, batch_size))
        epoch += 1


def load_interp_batches(tfrecord_path, batch_size, image_size=(384, 512), seed=0, epochs=None, scale=False,
                        shuffle_buffer=256):
    """Batches for FlowNetS_interp from the reference's `image_matches` records (dataloader.py:211-245: image_a and
    matches_a float64, sparse_flow / edges_a / flow float32, all at the dataset's padded size): device tensors
    (image_a [B,h,w,3], matches_a [B,h,w,1], sparse_flow [B,h,w,2], edges_a [B,h,w,1], flow [B,h,w,2]).  No
    augmentation: the reference's `augment_all_interp` (random crops / flips / colour / re-sampling of the sparse
    flow, dataloader.py:30-80) is not built."""
    from . import tfrecord
    names = ("image_a", "matches_a", "sparse_flow", "edges_a", "flow")
    rng = np.random.default_rng(seed)
    div = 255.0 if scale else 1.0
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(P._hip.require_device())
    epoch = 0
    while epochs is None or epoch < epochs:
        buf, pick, produced = [], [], False

        def emit(items):
            return tuple(dev(np.stack([it[i] for it in items]).astype(np.float32)) for i in range(5))

        def samples():
            for smp in tfrecord.read_samples(tfrecord_path, image_size[0], image_size[1], names):
                item = (smp["image_a"] / div, smp["matches_a"] / div, smp["sparse_flow"], smp["edges_a"], smp["flow"])
                if len(buf) < shuffle_buffer:
                    buf.append(item)
                    continue
                k = int(rng.integers(len(buf)))
                out, buf[k] = buf[k], item
                yield out
            for k in rng.permutation(len(buf)):
                yield buf[k]

        for item in samples():
            pick.append(item)
            if len(pick) == batch_size:
                yield emit(pick)
                pick, produced = [], True
        if not produced:
            raise ValueError("%s holds fewer than one batch (%d) of samples" % (tfrecord_path, batch_size))
        epoch += 1
