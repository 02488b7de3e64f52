"""Generate tests/golden/flowlib_golden.npz by RUNNING the reference's own
src/flowlib.py (build container only: needs /root/reference).

The reference module imports two third-party packages that are not installed
here (``png``, ``imageio``); they are only used by KITTI-png helpers that this
script never calls, so empty placeholder modules are registered for the import.
Nothing of the reference's source is copied: only inputs and its outputs are
stored.

    python tests/golden/make_golden_flowlib.py
"""
import hashlib
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    for m in ("png", "imageio"):
        mod = types.ModuleType(m)
        mod.imread = lambda *a, **k: None
        sys.modules[m] = mod
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    from src import flowlib  # the reference implementation

    out = {}
    # (1) the reference's sample .flo files: read_flow stats + colour-coding digest
    for i in (0, 1):
        f = flowlib.read_flow(os.path.join(REF, "data/samples/%dflow.flo" % i))
        out["sample%d_shape" % i] = np.array(f.shape)
        out["sample%d_absmean" % i] = np.array(np.abs(f.astype(np.float64)).mean())
        out["sample%d_probe" % i] = f[::37, ::41].copy()
        img = flowlib.flow_to_image(f.copy())
        out["sample%d_viz_sha256" % i] = np.frombuffer(
            hashlib.sha256(np.ascontiguousarray(img).tobytes()).digest(), np.uint8)
        out["sample%d_viz_probe" % i] = img[::37, ::41].copy()
        img2 = flowlib.flow_to_image(f.copy(), maxflow=12.5)
        out["sample%d_viz_max12p5_probe" % i] = img2[::37, ::41].copy()
    # (2) a small synthetic field with unknown (>1e9) and NaN entries: full outputs
    rng = np.random.default_rng(7)
    f = (rng.standard_normal((24, 32, 2)) * 6).astype(np.float32)
    f[3, 4] = 2e9
    f[10, 11, 0] = np.nan
    f[0, 0] = 0
    out["synth_flow"] = f.copy()
    out["synth_viz"] = flowlib.flow_to_image(f.copy())
    out["synth_viz_max5"] = flowlib.flow_to_image(f.copy(), maxflow=5.0)
    out["zero_viz"] = flowlib.flow_to_image(np.zeros((4, 5, 2), np.float32))
    out["color_wheel"] = flowlib.make_color_wheel()
    # (3) write_flow byte image and read_flow round trip
    g = (rng.standard_normal((5, 7, 2)) * 3).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "x.flo")
        flowlib.write_flow(g, p)
        out["rt_flow"] = g
        out["rt_bytes"] = np.frombuffer(open(p, "rb").read(), np.uint8)
        out["rt_read"] = flowlib.read_flow(p)
    np.savez_compressed(os.path.join(HERE, "flowlib_golden.npz"), **out)
    print("wrote flowlib_golden.npz with", sorted(out))


if __name__ == "__main__":
    main()
