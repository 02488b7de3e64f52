"""Training-input augmentation behind the op surface of the reference's preprocessing plugin
(src/ops/preprocessing/preprocessing.cc:24-95: ops `DataAugmentation`, `FlowAugmentation`; the plugin is loaded
at src/dataloader.py:14-15).  Host side = what the op does in host memory: draw the coefficients
(augmentation_base.cc:190-300), re-draw until the crop's corners fit the source (:300-360), compose the 2x3
matrices (:8-60); device side = the two per-pixel HIP passes (csrc/aug.hip).  Random numbers come from a NumPy
Generator (the reference seeds std::mt19937 from std::random_device on every draw: not reproducible by design).
"""
import math

import numpy as np
import torch

from . import _hip

SPATIAL = ("translate", "rotate", "zoom", "squeeze")
CHROMATIC = ("gamma", "brightness", "contrast", "color")
_DEFAULTS = dict(dx=0.0, dy=0.0, angle=0.0, zoom_x=1.0, zoom_y=1.0, gamma=1.0, brightness=0.0, contrast=1.0,
                 color1=1.0, color2=1.0, color3=1.0)


def _params(names, rand_type, exp, mean, spread, prob):
    """The parallel attribute lists of the op -> {name: param}; 'noise' is handled on the Python side in the
    reference and ignored here too; unknown names are reported and skipped (augmentation_base.h:118-146)."""
    out = {}
    for i, name in enumerate(names):
        if name in SPATIAL + CHROMATIC:
            out[name] = dict(rand_type=rand_type[i], exp=bool(exp[i]), mean=float(mean[i]), spread=float(spread[i]),
                             prob=float(prob[i]))
        elif name != "noise":
            print("Ignoring unknown augmentation parameter: " + name)
    return out


def rng_generate(rng, param, discount_coeff, default_value):
    """augmentation_base.cc:190-250: Bernoulli(prob) gate, then uniform(mean +- spread*discount) or
    normal(mean, spread*discount), optionally exponentiated."""
    spread = param["spread"] * discount_coeff
    if param["rand_type"] not in ("uniform_bernoulli", "gaussian_bernoulli"):
        raise ValueError("Unknown random type: " + param["rand_type"])
    if not (param["prob"] > 0.0 and rng.random() < param["prob"]):
        return default_value
    if param["rand_type"] == "uniform_bernoulli":
        v = rng.uniform(param["mean"] - spread, param["mean"] + spread) if param["spread"] > 0.0 else param["mean"]
    else:
        v = rng.normal(param["mean"], spread) if spread > 0.0 else param["mean"]
    return math.exp(v) if param["exp"] else v


def generate_spatial_coeffs(rng, aug, discount):
    c = {}
    if "translate" in aug:
        c["dx"] = rng_generate(rng, aug["translate"], discount, _DEFAULTS["dx"])
        c["dy"] = rng_generate(rng, aug["translate"], discount, _DEFAULTS["dy"])
    if "rotate" in aug:
        c["angle"] = rng_generate(rng, aug["rotate"], discount, _DEFAULTS["angle"])
    if "zoom" in aug:
        c["zoom_x"] = rng_generate(rng, aug["zoom"], discount, _DEFAULTS["zoom_x"])
        c["zoom_y"] = c["zoom_x"]
    if "squeeze" in aug:
        s = rng_generate(rng, aug["squeeze"], discount, 1.0)
        c["zoom_x"] = c.get("zoom_x", 1.0) * s
        c["zoom_y"] = c.get("zoom_y", 1.0) * s  # the reference multiplies both by s (augmentation_base.cc:292-296)
    return c


def generate_chromatic_coeffs(rng, aug, discount, coeff):
    for name in ("gamma", "brightness", "contrast"):
        if name in aug:
            coeff[name] = rng_generate(rng, aug[name], discount, _DEFAULTS[name])
    if "color" in aug:
        for k in ("color1", "color2", "color3"):
            coeff[k] = rng_generate(rng, aug["color"], discount, _DEFAULTS[k])


def _combine(coeff, incoming):
    """AugmentationCoeff::combine_with (:107-153): every coefficient the incoming set HAS multiplies this one."""
    for k, v in incoming.items():
        coeff[k] = coeff.get(k, _DEFAULTS[k]) * v
    return coeff


def corners_fit(coeff, src_w, src_h, out_w, out_h):
    ang = coeff.get("angle", 0.0)
    for x in (0, out_w - 1):
        for y in (0, out_h - 1):
            x1, y1 = x - 0.5 * out_w, y - 0.5 * out_h
            x2 = (math.cos(ang) * x1 - math.sin(ang) * y1 + coeff.get("dx", 0.0) * out_w) / coeff.get("zoom_x", 1.0)
            y2 = (math.sin(ang) * x1 + math.cos(ang) * y1 + coeff.get("dy", 0.0) * out_h) / coeff.get("zoom_y", 1.0)
            x2, y2 = x2 + 0.5 * src_w, y2 + 0.5 * src_h
            if math.floor(x2) < 0 or math.floor(x2) > src_w - 2.0 or math.floor(y2) < 0 or math.floor(y2) > src_h - 2.0:
                return False
    return True


def generate_valid_spatial_coeffs(rng, aug, discount, incoming, src_w, src_h, out_w, out_h):
    """Up to 50 draws until all four corners of the crop fall inside the source (:300-360); afterwards the
    incoming coefficients alone."""
    for _ in range(50):
        c = _combine(generate_spatial_coeffs(rng, aug, discount), incoming)
        if corners_fit(c, src_w, src_h, out_w, out_h):
            return c
    print("Warning: No suitable spatial transformation after 50 attempts.")
    return dict(incoming)


def _left_multiply(t, u):
    t0, t1, t2, t3, t4, t5 = t
    u0, u1, u2, u3, u4, u5 = (np.float32(v) for v in u)
    return [t0 * u0 + t3 * u1, t1 * u0 + t4 * u1, t2 * u0 + t5 * u1 + u2,
            t0 * u3 + t3 * u4, t1 * u3 + t4 * u4, t2 * u3 + t5 * u4 + u5]


def transmat_from_coeff(coeff, out_w, out_h, src_w, src_h):
    """TransMat::fromCoeff (:8-35) in float32: output pixel -> source position."""
    f = np.float32
    t = [f(1), f(0), f(0), f(0), f(1), f(0)]
    t = _left_multiply(t, (1, 0, -0.5 * out_w, 0, 1, -0.5 * out_h))
    if "angle" in coeff:
        a = coeff["angle"]
        t = _left_multiply(t, (math.cos(a), -math.sin(a), 0, math.sin(a), math.cos(a), 0))
    if "dx" in coeff or "dy" in coeff:
        t = _left_multiply(t, (1, 0, coeff.get("dx", 0.0) * out_w, 0, 1, coeff.get("dy", 0.0) * out_h))
    if "zoom_x" in coeff or "zoom_y" in coeff:
        t = _left_multiply(t, (1.0 / coeff.get("zoom_x", 1.0), 0, 0, 0, 1.0 / coeff.get("zoom_y", 1.0), 0))
    t = _left_multiply(t, (1, 0, 0.5 * src_w, 0, 1, 0.5 * src_h))
    return np.array(t, np.float32)


def transmat_inverse(t):
    a, b, c, d, e, f = (np.float32(v) for v in t)
    den = a * e - b * d
    return np.array([e / den, b / -den, (c * e - b * f) / -den, d / -den, a / den, (c * d - a * f) / den], np.float32)


def _discount(schedule, global_step):
    half_life, initial, final = schedule
    return initial + (final - initial) * (2.0 / (1.0 + math.exp(-1.0986 * global_step / half_life)) - 1.0)


def _dev(x):
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    return t.to(device=_hip.require_device(), dtype=torch.float32).contiguous()


def augment(images, transforms, chromatic, crop):
    """The device pass of DataAugmentation for explicit matrices / chromatic coefficients ([N,6] each)."""
    img, tr = _dev(images), _dev(transforms)
    if img.ndim != 4 or tr.shape != (img.shape[0], 6):
        raise ValueError("augment: images must be [N,H,W,C] and transforms [N,6]")
    ch = _dev(chromatic) if chromatic is not None else None
    n, h, w, c = img.shape
    out = torch.empty((n, int(crop[0]), int(crop[1]), c), dtype=torch.float32, device=img.device)
    _hip.check(_hip.lib().fn2_augment_f32(_hip.ptr(img), _hip.ptr(tr), _hip.ptr(ch) if ch is not None else None,
                                          _hip.ptr(out), n, h, w, c, int(crop[0]), int(crop[1]), _hip.stream_ptr()))
    return out


def data_augmentation(image_a, image_b, global_step, crop, params_a_name, params_a_rand_type, params_a_exp,
                      params_a_mean, params_a_spread, params_a_prob, params_a_coeff_schedule, params_b_name,
                      params_b_rand_type, params_b_exp, params_b_mean, params_b_spread, params_b_prob,
                      params_b_coeff_schedule, seed=None):
    """`_preprocessing_ops.data_augmentation(...)`: returns (aug_image_a, aug_image_b, transforms_from_a,
    transforms_from_b) -- transforms_from_b already INVERTED, ready for flow_augmentation
    (data_augmentation.cc:177-420).  Image b inherits image a's coefficients and multiplies its own on top."""
    if len(crop) != 2:
        raise ValueError("crop must be 2 dimensions")
    a, b = _dev(image_a), _dev(image_b)
    if a.ndim != 4 or a.shape != b.shape:
        raise ValueError("image_a and image_b must be rank 4 and of equal shape")
    n, sh, sw, _ = a.shape
    oh, ow = int(crop[0]), int(crop[1])
    rng = np.random.default_rng(seed)
    aug_a = _params(params_a_name, params_a_rand_type, params_a_exp, params_a_mean, params_a_spread, params_a_prob)
    aug_b = _params(params_b_name, params_b_rand_type, params_b_exp, params_b_mean, params_b_spread, params_b_prob)
    disc_a = _discount(params_a_coeff_schedule, int(global_step)) if len(params_a_coeff_schedule) == 3 else 1.0
    disc_b = 1.0
    if len(params_b_coeff_schedule) == 3:
        disc_b = disc_a if len(params_a_coeff_schedule) == 3 else _discount(params_b_coeff_schedule, int(global_step))
    spatial_a, chroma_a = any(k in aug_a for k in SPATIAL), any(k in aug_a for k in CHROMATIC)
    spatial_b, chroma_b = any(k in aug_b for k in SPATIAL), any(k in aug_b for k in CHROMATIC)
    coeffs_a, coeffs_b = [], []
    for _ in range(n):
        c = generate_valid_spatial_coeffs(rng, aug_a, disc_a, {}, sw, sh, ow, oh) if spatial_a else {}
        if chroma_a:
            generate_chromatic_coeffs(rng, aug_a, disc_a, c)
        coeffs_a.append(c)
    for ca in coeffs_a:
        c = dict(ca)
        if spatial_b:
            # the reference clears the whole coefficient set, draws b's spatial ones and MULTIPLIES image a's set
            # back in (combine_with, :107-153): dx_b * dx_a, angle_b * angle_a, ... and, since a cleared brightness
            # reads as 0, image a's brightness offset is lost for image b (0 * brightness_a)
            c = generate_valid_spatial_coeffs(rng, aug_b, disc_b, ca, sw, sh, ow, oh)
        if chroma_b:
            generate_chromatic_coeffs(rng, aug_b, disc_b, c)
        coeffs_b.append(c)
    ta = np.stack([transmat_from_coeff(c, ow, oh, sw, sh) for c in coeffs_a])
    tb = np.stack([transmat_from_coeff(c, ow, oh, sw, sh) for c in coeffs_b])

    def chroma_table(coeffs):
        return np.array([[c.get(k, _DEFAULTS[k]) for k in ("gamma", "brightness", "contrast", "color1", "color2", "color3")]
                         for c in coeffs], np.float32)

    out_a = augment(a, ta, chroma_table(coeffs_a) if chroma_a else None, crop)
    out_b = augment(b, tb, chroma_table(coeffs_b) if (chroma_a or chroma_b) else None, crop)
    inv_tb = np.stack([transmat_inverse(t) for t in tb])
    return out_a, out_b, torch.from_numpy(ta), torch.from_numpy(inv_tb)


def flow_augmentation(flows, transforms_from_a, transforms_from_b, crop):
    """`_preprocessing_ops.flow_augmentation(flows, transforms_from_a, transforms_from_b, crop)`: the ground-truth
    flow seen through the two augmentations (flow_augmentation.cc:69-125; same argument checks)."""
    if len(crop) != 2:
        raise ValueError("crop must be 2 dimensions")
    f = _dev(flows)
    if f.ndim != 4:
        raise ValueError("Input images must have rank 4")
    ta, tb = _dev(transforms_from_a), _dev(transforms_from_b)
    if tuple(ta.shape) != (f.shape[0], 6):
        raise ValueError("Input transforms_from_a should be num_images x 6")
    if tuple(tb.shape) != (f.shape[0], 6):
        raise ValueError("Input transforms_from_b should be num_images x 6")
    n, h, w, _ = f.shape
    out = torch.empty((n, int(crop[0]), int(crop[1]), f.shape[3]), dtype=torch.float32, device=f.device)
    _hip.check(_hip.lib().fn2_flow_augmentation_f32(_hip.ptr(f), _hip.ptr(ta), _hip.ptr(tb), _hip.ptr(out), n, h, w,
                                                    int(crop[0]), int(crop[1]), _hip.stream_ptr()))
    return out
