"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header
declares, argument validation that needs no device, weight packing, harness logic."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    from src import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "flownet2-tf_amd", "csrc"), "-j4"], check=True)
    return _hip


def test_library_exports_every_declared_symbol(hip):
    header = open(os.path.join(ROOT, "include", "flownet2_hip.h")).read()
    declared = set(re.findall(r"\b(fn2_[a-z0-9_]+)\s*\(", header))
    declared -= {"fn2_pack_"}  # prose in a comment
    assert len(declared) >= 20
    lib = C.CDLL(hip.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    # and the ctypes prototype table covers the header exactly
    assert declared == set(hip.PROTOTYPES), declared ^ set(hip.PROTOTYPES)


def test_correlation_out_shape_and_validation_without_gpu(hip):
    lib = hip.lib()
    oh, ow, oc = C.c_int(), C.c_int(), C.c_int()
    assert lib.fn2_correlation_out_shape(48, 64, 1, 20, 1, 2, 20, C.byref(oh), C.byref(ow), C.byref(oc)) == 0
    assert (oh.value, ow.value, oc.value) == (48, 64, 441)  # flownet_c.py:40 at 512x384
    assert lib.fn2_correlation_out_shape(56, 128, 1, 20, 1, 2, 20, C.byref(oh), C.byref(ow), C.byref(oc)) == 0
    assert (oh.value, ow.value, oc.value) == (56, 128, 441)  # 1024x448
    assert lib.fn2_correlation_out_shape(12, 14, 3, 2, 2, 1, 3, C.byref(oh), C.byref(ow), C.byref(oc)) == 0
    from oracle import ops
    g = ops.correlation_geometry(12, 14, 3, 2, 2, 1, 3)
    assert (oh.value, ow.value, oc.value) == (g["oh"], g["ow"], g["D"])
    rc = lib.fn2_correlation_out_shape(8, 8, 2, 2, 1, 1, 2, C.byref(oh), C.byref(ow), C.byref(oc))
    assert rc == hip.ERR_INVALID_ARGUMENT and b"odd" in lib.fn2_last_error()
    with pytest.raises(ValueError):
        hip.check(rc)
    rc = lib.fn2_correlation_out_shape(4, 4, 1, 8, 1, 1, 0, C.byref(oh), C.byref(ow), C.byref(oc))
    assert rc == hip.ERR_INVALID_ARGUMENT and b"fit" in lib.fn2_last_error()


def test_conv_plan_rules(hip):
    def plan(code, cin_pad, cout):
        p = hip.conv_plan(code, cin_pad, cout)
        return (p.layout, p.cout_tile, p.kstep_elems, p.wgt_dtype)
    assert plan(hip.FN2_BF16, 64, 128) == (1, 128, 64, hip.FN2_BF16)   # whole 128-byte lines: LDS-DMA kernel
    assert plan(hip.FN2_BF16, 32, 128) == (0, 128, 32, hip.FN2_BF16)   # 64 bytes per tap: generic kernel
    assert plan(hip.FN2_F32, 32, 64) == (1, 64, 32, hip.FN2_F32)
    assert plan(hip.FN2_F32, 8, 64) == (0, 64, 16, hip.FN2_F32)        # stem
    assert plan(hip.FN2_F16, 256, 32) == (1, 32, 64, hip.FN2_F16)      # Cout <= 32: the 32-cout x 256-pixel tile
    assert plan(hip.FN2_F16, 256, 48) == (1, 64, 64, hip.FN2_F16)
    assert plan(hip.FN2_BF16, 256, 2) == (0, 16, 32, hip.FN2_BF16)     # flow head
    assert plan(hip.FN2_F16X2, 256, 2) == (0, 16, 16, hip.FN2_F32)     # flow head on split fp16: fp32 weights
    assert plan(hip.FN2_F16X2, 96, 16) == (1, 32, 32, hip.FN2_F16X2)
    with pytest.raises(NotImplementedError):
        hip.conv_plan(hip.FN2_F16X2, 8, 64)                            # split fp16 has no generic input path
    with pytest.raises(ValueError):
        hip.conv_plan(hip.FN2_F32, 12, 64)


def test_split_fp16_roundtrip():
    from src import weights as W
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((5, 7, 24)) * np.exp(rng.uniform(-6, 3, (5, 7, 24)))).astype(np.float32)
    s = W.split_f16x2(x)
    assert s.dtype == np.float16 and s.shape == (5, 7, 48)
    np.testing.assert_array_equal(s[..., :8].astype(np.float32), x[..., :8].astype(np.float16).astype(np.float32))
    back = W.join_f16x2(s)
    err = np.abs(back - x)
    big = np.abs(x) >= 0.125   # lo part is a normal fp16 number: 22 significant bits
    assert np.max(err[big] / np.abs(x)[big]) < 2.0 ** -21
    assert np.max(err[~big]) <= 2.0 ** -24  # lo part subnormal: absolute error half an fp16 subnormal step


def test_pack_conv_layouts():
    from src import weights as W
    rng = np.random.default_rng(0)
    w = rng.standard_normal((3, 3, 5, 70)).astype(np.float32)
    p, cin_pad, cout_pad, kpad = W.pack_conv(w, 128, 32)
    assert (cin_pad, cout_pad, kpad) == (8, 128, 96) and p.shape == (128, 96)
    assert p[7, (1 * 3 + 2) * 8 + 4] == w[1, 2, 4, 7]
    assert np.all(p[70:] == 0) and np.all(p[:, 72:] == 0) and np.all(p.reshape(128, -1)[:, 5:8] == 0)
    q, _, _, _ = W.pack_conv(w, 128, 32, cin_pad=64, layout=1)
    assert q.shape == (128, 9 * 64)
    # layout 1: inside each group of 32 rows, packed row (r&3) + 8*(r>>2) + 4*h <- cout 16*h + r
    for (h, r) in [(0, 0), (1, 0), (0, 5), (1, 15), (0, 10)]:
        assert np.array_equal(q[(r & 3) + 8 * (r >> 2) + 4 * h, :5], w[0, 0, :, 16 * h + r])
    assert np.array_equal(q[64 + (5 & 3) + 8 * (5 >> 2) + 4 * 0, :5], w[0, 0, :, 64 + 5])


def test_pack_deconv_phases_reproduce_transposed_conv():
    """Pure-NumPy check of the phase decomposition used by the kernel (kind 1)."""
    from oracle import nn as refnn
    from src import weights as W
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1, 3, 4, 5))
    w = rng.standard_normal((4, 4, 6, 5))
    want = refnn.conv2d_transpose(x, w)
    p, cin_pad, cout_pad, kpad = W.pack_deconv(w.astype(np.float32), 16, 32)
    got = np.zeros_like(want)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    for a in range(2):
        for b in range(2):
            wp = p[a * 2 + b][:6, :4 * cin_pad].reshape(6, 2, 2, cin_pad)[..., :5]
            for y in range(3):
                for xx in range(4):
                    acc = np.zeros(6)
                    for ty in range(2):
                        for tx in range(2):
                            acc += wp[:, ty, tx] @ xp[0, y + a + ty, xx + b + tx]  # iy = y-1+a+ty (+1 pad)
                    got[0, 2 * y + a, 2 * xx + b] = acc
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_weights_roundtrip_and_names(tmp_path):
    from src import netdefs, weights as W
    w = W.init_weights("FlowNet2", 1)
    assert "FlowNet2/FlowNetCSS/FlowNetCS/FlowNetC/conv1/weights" in w  # SURVEY.md A.6 scope nesting
    assert w["FlowNet2/FlowNetCSS/FlowNetS/conv1/weights"].shape == (7, 7, 12, 64)
    assert w["FlowNet2/FlowNetSD/deconv5/weights"].shape == (4, 4, 512, 1024)  # HW-O-I
    assert "FlowNet2/FlowNetSD/deconv5/biases" not in w  # biases_initializer=None
    assert w["FlowNet2/fuse_interconv0/weights"].shape == (3, 3, 82, 16)
    n_params = sum(v.size for k, v in w.items() if k.endswith("weights"))
    assert abs(n_params / 1e6 - 162.5) < 2.0  # ~163.5 M incl. biases
    small = W.init_weights("FlowNetS", 2)
    W.save_npz(tmp_path / "s.npz", small)
    back = W.load_npz(tmp_path / "s.npz")
    assert set(back) == set(small) and all(np.array_equal(back[k], small[k]) for k in small)
    assert sum(v.size for v in small.values()) / 1e6 == pytest.approx(38.68, abs=0.05)
    for m in netdefs.MODELS:
        assert netdefs.model_scopes(m)


def test_adapt_x_pad_crop_and_flo_output(tmp_path, golden_dir):
    from src.net import Net, imread
    from src import flowlib
    net = Net()
    a = (np.arange(436 * 1024 * 3) % 251).reshape(436, 1024, 3).astype(np.uint8)
    a1, b1, info = net.adapt_x(a, a)
    assert a1.shape == (1, 448, 1024, 3) and info == (1, 436, 1024, 3)  # Sintel -> 448 (net.py:373-388)
    assert a1.dtype == np.float32 and a1.max() <= 1.0 and np.all(a1[:, 436:] == 0)
    flow = np.zeros((448, 1024, 2), np.float32)
    assert net.postproc_y_hat_test(flow, (436, 1024, 2)).shape == (436, 1024, 2)
    assert net.get_padded_image_size(384, 512) == (384, 512)
    img = imread(os.path.join(golden_dir, "samples", "0img0.ppm"))
    assert img.shape == (384, 512, 3) and img.dtype == np.uint8
    a2, _, info2 = net.adapt_x(img, img)
    assert info2 is None and a2.shape == (1, 384, 512, 3)
    with pytest.raises(AssertionError):
        net.adapt_x(img, img[:100])
    # product flowlib == reference bytes / colours (same goldens as the oracle)
    g = np.load(os.path.join(golden_dir, "flowlib_golden.npz"))
    p = tmp_path / "x.flo"
    flowlib.write_flow(g["rt_flow"], p)
    assert open(p, "rb").read() == g["rt_bytes"].tobytes()
    assert np.array_equal(flowlib.read_flow(p), g["rt_read"])
    assert np.array_equal(flowlib.make_color_wheel(), g["color_wheel"])
    assert np.array_equal(flowlib.flow_to_image(g["synth_flow"].copy()), g["synth_viz"])
    assert np.array_equal(flowlib.flow_to_image(g["synth_flow"].copy(), maxflow=5.0), g["synth_viz_max5"])


def test_ops_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from src.correlation import correlation
    from src.flow_warp import flow_warp
    from src.downsample import downsample
    x = np.zeros((1, 4, 4, 2), np.float32)
    for call in (lambda: correlation(x, x, 1, 2, 1, 1, 2), lambda: flow_warp(x, x), lambda: downsample(x, [2, 2])):
        with pytest.raises(RuntimeError, match="no ROCm device"):
            call()


def test_cli_argument_checks(tmp_path):
    import subprocess
    import sys
    pkg = os.path.join(ROOT, "flownet2-tf_amd")
    r = subprocess.run([sys.executable, "-m", "src.flownet_s.test", "--input_a", "/nonexistent.png", "--input_b",
                        "/nonexistent.png", "--out", str(tmp_path)], cwd=pkg, capture_output=True, text=True)
    assert r.returncode != 0 and "image_a path must exist" in r.stderr
    r = subprocess.run([sys.executable, "-m", "src.flownet2.test", "--input_a", os.path.join(ROOT, "bench.py")],
                       cwd=pkg, capture_output=True, text=True)
    assert r.returncode != 0 and "required" in r.stderr


def test_hfem_losses_match_oracle_on_cpu_tensors():
    """src.losses.average_endpoint_error_hfem (the mining modes: a torch top-k selection + sums) on CPU tensors against
    the NumPy restatement of utils.py:227-339.  (average_endpoint_error / mean_endpoint_error are calls into the HIP
    library and are checked on the GPU: tests/test_gpu_models.py, tests/test_gpu_train.py.)"""
    import torch
    from oracle import models as refm
    from src import losses
    rng = np.random.default_rng(5)
    lab, pred = rng.standard_normal((3, 6, 7, 2)).astype(np.float32), rng.standard_normal((3, 6, 7, 2)).astype(np.float32)
    edges = rng.random((3, 6, 7, 1)).astype(np.float32)
    tl, tp, te = torch.from_numpy(lab), torch.from_numpy(pred), torch.from_numpy(edges)
    for mode, kw in (("", {}), ("hard", {}), ("hard", {"perc_hfem": 33, "lambda_w": 1.0}), ("edges", {"edges": edges}),
                     ("edges", {}), ("other", {})):
        tkw = {k: (te if k == "edges" else v) for k, v in kw.items()}
        got = float(losses.average_endpoint_error_hfem(tl, tp, mode, **tkw))
        assert got == pytest.approx(refm.average_endpoint_error_hfem(lab, pred, mode, **kw), rel=1e-5)


def test_interp_weights_and_class_surface():
    from src import weights as W
    from src.flownet_s_interp.flownet_s_interp import FlowNetS_interp
    w = W.init_weights("FlowNetS_interp", 3)
    assert "FlowNetS/conv1/weights" in w and w["FlowNetS/conv1/weights"].shape == (7, 7, 6, 64)
    assert not any("/predict_flow" in k and k.endswith("/biases") for k in w)   # no_deconv_biases default
    net = FlowNetS_interp()
    assert net.scope == "FlowNetS" and net.no_deconv_biases is True
    a, m, sf, info = net.adapt_x_matches(np.full((436, 1024, 3), 200, np.uint8), np.full((436, 1024), 255, np.uint8),
                                         np.ones((436, 1024, 2), np.float32))
    assert a.shape == (1, 448, 1024, 3) and m.shape == (1, 448, 1024, 1) and sf.shape == (1, 448, 1024, 2)
    assert info == (1, 436, 1024, 3) and m.max() == 1.0 and a.max() == pytest.approx(200 / 255.0)
    assert sf[0, 440].max() == 0.0


def test_caffe_converter_npy_is_ingested(tmp_path):
    """scripts/caffe/convert_caffe_weights_to_npy.py:489-496 writes np.save(dict): same names and layouts."""
    from src import weights as W
    from src.net import Net
    w = W.init_weights("FlowNetS", 4)
    extra = dict(w)
    extra["FlowNetS/deconv5/biases"] = np.zeros(512, np.float32)      # Caffe has them, the TF graph does not read them
    np.save(tmp_path / "flownet_s.npy", extra)
    back = W.load_weights(str(tmp_path / "flownet_s.npy"))
    assert set(back) == set(extra) and all(np.array_equal(back[k], extra[k]) for k in extra)

    class S(Net):
        model_name = "FlowNetS"
    n = S()
    assert set(n.load_weights(str(tmp_path / "flownet_s.npy"))) == set(extra)
    np.save(tmp_path / "bad.npy", np.zeros(3))
    with pytest.raises(ValueError):
        W.load_npy(str(tmp_path / "bad.npy"))


def test_bench_traffic_lookup_resolves_committed_profiles():
    """bench.py's roofline.traffic comes from the committed PMC summaries: the kernel names the engine reports must
    resolve in them (a renamed instantiation silently turned the field into null once)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    for model, batch in (("FlowNetC", 8), ("FlowNet2", 4)):
        path = os.path.join(ROOT, "profiles", "pmc_traffic_%s_b%d_f16x2.json" % (model, batch))
        if not os.path.exists(path):
            continue
        # the newest round's committed bench line of this workload names the kernel the PMC pass was keyed on
        import glob
        lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*%s_b%d_f16x2_bench.json" % (model.lower(), batch))),
                       key=os.path.basename)
        line = json.load(open(lines[-1]))
        kern = line["roofline"]["kernel"]
        got = bench.stored_traffic(model, batch, "f16x2", kern)
        assert got is not None and got > 1e6, (model, kern)
    assert bench.stored_traffic("FlowNetC", 8, "f16x2", "no_such_kernel<1>") is None


def test_bias_declarations_follow_the_reference_arg_scopes():
    """netdefs.has_bias: slim's default (a zeros-initialised `biases` variable) unless a scope says
    biases_initializer=None -- flownet_s.py:53, flownet_c.py:58, flownet_sd.py:44 do for conv2d_transpose, the FlowNet2
    fusion net (flownet2.py:50-57) does not, FlowNetS_interp decides per constructor flag (flownet_s_interp.py:78-126)."""
    from src import netdefs, weights as W
    for model in ("FlowNetS", "FlowNetC", "FlowNetSD", "FlowNetCS", "FlowNetCSS"):
        keys = W.init_weights(model, 1)
        assert not any(("deconv" in k or "upsample_flow" in k) and k.endswith("/biases") for k in keys), model
        assert all(k[:-len("weights")] + "biases" in keys for k in keys
                   if k.endswith("/weights") and "deconv" not in k and "upsample_flow" not in k), model
    w2 = W.init_weights("FlowNet2", 1)
    fused = sorted(k for k in w2 if ("deconv" in k or "upsample_flow" in k) and k.endswith("/biases"))
    assert fused == ["FlowNet2/fuse_deconv0/biases", "FlowNet2/fuse_deconv1/biases",
                     "FlowNet2/fuse_upsample_flow1to0/biases", "FlowNet2/fuse_upsample_flow2to1/biases"]
    assert w2["FlowNet2/fuse_deconv1/biases"].shape == (32,) and w2["FlowNet2/fuse_deconv0/biases"].shape == (16,)
    assert all(np.abs(w2[k]).max() > 0 for k in fused)  # non-zero: a path that drops them fails its parity test
    # the sub-networks inside FlowNet2 keep their bias-free transposed convs
    assert "FlowNet2/FlowNetSD/deconv5/biases" not in w2 and "FlowNet2/FlowNetCSS/FlowNetS/deconv2/biases" not in w2
    wi = W.init_weights("FlowNetS_interp", 1)                       # no_deconv_biases=True (the class default)
    assert not any(("predict_flow" in k or "deconv" in k) and k.endswith("/biases") for k in wi)
    wb = W.init_weights("FlowNetS_interp", 1, head_biases=True)     # no_deconv_biases=False
    assert "FlowNetS/predict_flow4/biases" in wb and "FlowNetS/deconv3/biases" in wb
    assert not any("upsample_flow" in k and k.endswith("/biases") for k in wb)
    assert netdefs.has_bias("FlowNet2", "fuse_upsample_flow2to1", "deconv") and not netdefs.has_bias("FlowNetS", "deconv5", "deconv")
    # the weights of every layer are what round-1 fixtures were generated with (bias draws do not shift the stream)
    assert np.array_equal(W.init_weights("FlowNet2", 1)["FlowNet2/predict_flow0/weights"], w2["FlowNet2/predict_flow0/weights"])


def test_oracle_fusion_transposed_convs_read_their_biases():
    """oracle.models.flownet2's fusion part (flownet2.py:61-98): fuse_deconv1/0 and fuse_upsample_flow2to1/1to0 add their
    biases (before the LeakyReLU, as slim does) and a checkpoint lacking one is an error, not a silent zero."""
    from oracle import nn as refnn, models as refm
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 3, 4, 5))
    w = rng.standard_normal((4, 4, 2, 5))
    b = np.array([0.5, -2.0])
    plain = refnn.conv2d_transpose(x, w)
    np.testing.assert_allclose(refnn.conv2d_transpose(x, w, bias=b), plain + b, rtol=0, atol=1e-12)
    act = refnn.conv2d_transpose(x, w, bias=b, activation=refnn.leaky_relu)
    np.testing.assert_allclose(act, refnn.leaky_relu(plain + b), rtol=0, atol=1e-12)   # bias first, then LeakyReLU
    sc = refm._Scope({"S/d/weights": w, "S/d/biases": b}, "S")
    np.testing.assert_allclose(sc.deconv(x, "d", act=False, bias=True), plain + b, atol=1e-12)
    np.testing.assert_allclose(sc.deconv(x, "d", act=False), plain, atol=1e-12)  # biases_initializer=None scope: not read
    with pytest.raises(KeyError):
        refm._Scope({"S/d/weights": w}, "S").deconv(x, "d", bias=True)
