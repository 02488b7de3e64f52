"""End-to-end parity: the HIP engine against the fp64 oracle of the reference graphs on the
same seeded weights and inputs.  Tolerance from BASELINE.json north_star: mean EPE < 1e-3 px
(fp32 MFMA path).  The bf16 path is reported against its own, looser, stated bound."""
import os

import numpy as np
import pytest
import torch

from oracle import models as refm

pytestmark = pytest.mark.gpu

EPE_TOL = 1e-3  # px, BASELINE.json north_star


def images(n, h, w, seed):
    """SURVEY.md section 8d synthetic pairs: uint8 noise image, second = first rolled by (3,-5) + noise."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (n, h, w, 3)).astype(np.float32)
    # smooth a little so that the correlation has structure
    a = (a + np.roll(a, 1, 1) + np.roll(a, 1, 2) + np.roll(a, (1, 1), (1, 2))) / 4
    b = np.clip(np.roll(a, (3, -5), (1, 2)) + rng.uniform(-4, 4, a.shape), 0, 255)
    return (a / 255).astype(np.float32), (b / 255).astype(np.float32)


def epe(x, y):
    d = np.asarray(x, np.float64) - np.asarray(y, np.float64)
    return float(np.sqrt((d * d).sum(-1)).mean())


def run(model, dtype, n, h, w, seed=0):
    from src import weights as W
    from src.engine import Engine
    wts = W.init_weights(model, 1234)
    a, b = images(n, h, w, seed)
    eng = Engine(model, wts, n, h, w, dtype)
    out = {k: v.float().cpu().numpy() for k, v in eng(a, b).items()}
    want = refm.MODELS[model](wts, {"input_a": a, "input_b": b})
    return out, want


@pytest.mark.parametrize("model", ["FlowNetS", "FlowNetC", "FlowNetSD"])
def test_single_nets_f32(model):
    out, want = run(model, "f32", 2, 64, 128)
    for k in ("predict_flow6", "predict_flow5", "predict_flow4", "predict_flow3", "predict_flow2"):
        np.testing.assert_allclose(out[k], want[k], rtol=1e-3, atol=2e-4, err_msg=k)
    e = epe(out["flow"], want["flow"])
    mag = float(np.sqrt((want["flow"] ** 2).sum(-1)).mean())
    print(model, "mean EPE vs oracle %.3e px (mean |flow| %.2f px)" % (e, mag))
    assert e < EPE_TOL


@pytest.mark.parametrize("model", ["FlowNetCS", "FlowNetCSS", "FlowNet2"])
def test_stacked_nets_f32(model):
    out, want = run(model, "f32", 1, 64, 64)
    e = epe(out["flow"], want["flow"])
    mag = float(np.sqrt((want["flow"] ** 2).sum(-1)).mean())
    print(model, "mean EPE vs oracle %.3e px (mean |flow| %.2f px)" % (e, mag))
    assert e < EPE_TOL


def test_flownet_s_sample_pair_f32(golden_dir):
    """BASELINE config 1: FlowNetS on data/samples/0img0.ppm + 0img1.ppm (512x384)."""
    from src import weights as W
    from src.net import imread
    from src.flownet_s.flownet_s import FlowNetS
    a = imread(os.path.join(golden_dir, "samples", "0img0.ppm"))
    b = imread(os.path.join(golden_dir, "samples", "0img1.ppm"))
    net = FlowNetS()
    a1, b1, info = net.adapt_x(a, b)
    assert info is None and a1.shape == (1, 384, 512, 3)
    net.load_weights(None, seed=1234)
    preds = net.model({"input_a": a1, "input_b": b1})
    want = refm.flownet_s(net.weights, {"input_a": a1, "input_b": b1})
    e = epe(preds["flow"].cpu().numpy(), want["flow"])
    print("FlowNetS 512x384 sample pair: mean EPE vs oracle %.3e px" % e)
    assert e < EPE_TOL


def test_flownet2_full_size_f32():
    """BASELINE config 3 shape (one pair of it): FlowNet2 full stack at 512x384, fp32 MFMA path."""
    out, want = run("FlowNet2", "f32", 1, 384, 512)
    e = epe(out["flow"], want["flow"])
    mag = float(np.sqrt((want["flow"] ** 2).sum(-1)).mean())
    print("FlowNet2 512x384 mean EPE vs oracle %.3e px (mean |flow| %.2f px)" % (e, mag))
    assert e < EPE_TOL


@pytest.mark.parametrize("model", ["FlowNetS", "FlowNetC"])
def test_bf16_path_bounded(model):
    """bf16 activations/weights: not the parity path.  Stated bound: mean EPE < 5% of the mean
    flow magnitude (bf16 has 8 mantissa bits; ~25 layers)."""
    out, want = run(model, "bf16", 2, 64, 128)
    e = epe(out["flow"], want["flow"])
    mag = float(np.sqrt((want["flow"] ** 2).sum(-1)).mean())
    print(model, "bf16 mean EPE vs oracle %.3e px (mean |flow| %.2f px)" % (e, mag))
    assert e < 0.05 * mag


@pytest.mark.parametrize("model", ["FlowNetS", "FlowNetC", "FlowNetSD", "FlowNet2"])
def test_split_fp16_path_meets_parity_bar(model):
    """f16x2: split-fp16 storage, 3 fp16 MFMAs per product, fp32 accumulate.  Same 1e-3 px bar as fp32."""
    n, h, w = (1, 64, 64) if model == "FlowNet2" else (2, 64, 128)
    out, want = run(model, "f16x2", n, h, w)
    e = epe(out["flow"], want["flow"])
    mag = float(np.sqrt((want["flow"] ** 2).sum(-1)).mean())
    print(model, "f16x2 mean EPE vs oracle %.3e px (mean |flow| %.2f px)" % (e, mag))
    assert e < EPE_TOL


@pytest.mark.parametrize("model", ["FlowNetCS", "FlowNetCSS"])
def test_stacked_nets_split_fp16(model):
    """CS / CSS (flownet_cs.py:15-38, flownet_css.py:15-38) in the bench dtype."""
    out, want = run(model, "f16x2", 1, 64, 64)
    e = epe(out["flow"], want["flow"])
    print(model, "f16x2 mean EPE vs oracle %.3e px" % e)
    assert e < EPE_TOL


_SUBFLOWS = {"flow_c": "F2/CSS/CS/C/flow", "flow_cs": "F2/CSS/CS/S/flow", "flow_css": "F2/CSS/S/flow", "flow_sd": "F2/SD/flow"}


@pytest.mark.parametrize("dtype", ["f32", "f16x2"])
@pytest.mark.parametrize("shape", ["384x512", "448x1024"])
def test_flownet2_full_size_against_the_committed_fixture(golden_dir, shape, dtype):
    """BASELINE configs 3 (512x384) and 5 (Sintel 1024x436 zero-padded to 448 rows, net.py:373-388): the FlowNet2 full
    stack in the fp32 path and in the bench dtype against oracle outputs committed as
    tests/golden/flownet2_<shape>_golden.npz (4096 probe pixels of the final flow and of the C / CS / CSS / SD flows;
    generated by tests/golden/make_golden_flownet2.py with the fusion biases non-zero).  No oracle run on the box."""
    import sys
    sys.path.insert(0, golden_dir)
    import make_golden_flownet2 as gen
    from src import weights as W
    from src.engine import Engine
    g = np.load(os.path.join(golden_dir, "flownet2_%s_golden.npz" % shape))
    a, b = gen.inputs(shape)
    H, Wd, _ = gen.SHAPES[shape]
    eng = Engine("FlowNet2", W.init_weights("FlowNet2", gen.SEED_W), 1, H, Wd, dtype)
    flow = eng(a, b)["flow"].float().cpu().numpy()
    ys, xs = g["probe_y"], g["probe_x"]
    for key, buf in _SUBFLOWS.items():
        e = epe(eng.bufs[buf].float().cpu().numpy()[0, ys, xs], g[key])
        print("FlowNet2 %s %s %s: mean EPE over the probes %.3e px" % (shape, dtype, key, e))
        assert e < EPE_TOL, key
    e = epe(flow[0, ys, xs], g["flow"])
    print("FlowNet2 %s %s final flow: mean EPE over the probes %.3e px (mean |flow| %.3f px)" % (shape, dtype, e, float(g["mean_mag"])))
    assert e < EPE_TOL
    assert np.isfinite(flow).all()


def test_flownet2_fusion_biases_are_applied():
    """The fusion net's four transposed convs carry biases (flownet2.py:50-89: no biases_initializer=None scope): with
    them zeroed the flow moves by far more than the parity tolerance, and a checkpoint lacking one is an error."""
    from src import weights as W
    from src.engine import Engine
    wts = W.init_weights("FlowNet2", 1234)
    a, b = images(1, 64, 64, 0)
    base = Engine("FlowNet2", wts, 1, 64, 64, "f32")(a, b)["flow"].float().cpu().numpy()
    fused = [k for k in wts if k.startswith("FlowNet2/fuse_") and ("deconv" in k or "upsample" in k) and k.endswith("/biases")]
    assert len(fused) == 4
    zeroed = dict(wts)
    for k in fused:
        zeroed[k] = np.zeros_like(wts[k])
    moved = Engine("FlowNet2", zeroed, 1, 64, 64, "f32")(a, b)["flow"].float().cpu().numpy()
    assert epe(base, moved) > 3 * EPE_TOL
    want = refm.flownet2(zeroed, {"input_a": a, "input_b": b})["flow"]
    assert epe(moved, want) < EPE_TOL
    missing = {k: v for k, v in wts.items() if k != "FlowNet2/fuse_upsample_flow1to0/biases"}
    with pytest.raises(KeyError):
        Engine("FlowNet2", missing, 1, 64, 64, "f32")


def test_engine_reports_variables_no_layer_reads():
    """A variable under the model's scope that no layer consumes is an error (silently dropped tensors are how the
    fusion biases went missing); Caffe's deconvolution biases of a converted FlowNetS .npy are NOT graph variables
    (biases_initializer=None, flownet_s.py:53): ignored with a warning, as the reference's Saver ignores them; Adam
    slots and counters of a training checkpoint pass."""
    from src import weights as W
    from src.engine import Engine
    wts = W.init_weights("FlowNetS", 3)
    a, b = images(1, 64, 64, 1)
    base = Engine("FlowNetS", wts, 1, 64, 64, "f32")(a, b)["flow"].clone()
    stray = dict(wts)
    stray["FlowNetS/conv7/weights"] = np.zeros((3, 3, 8, 8), np.float32)
    with pytest.raises(ValueError, match="conv7"):
        Engine("FlowNetS", stray, 1, 64, 64, "f32")
    assert Engine("FlowNetS", stray, 1, 64, 64, "f32", strict=False).stray_variables == ["FlowNetS/conv7/weights"]
    caffe = dict(wts)
    caffe["FlowNetS/deconv5/biases"] = np.ones(512, np.float32)
    caffe["FlowNetS/upsample_flow6to5/biases"] = np.ones(2, np.float32)
    caffe["FlowNetS/conv1/weights/Adam"] = np.zeros((7, 7, 6, 64), np.float32)
    caffe["global_step"] = np.int64(7)
    with pytest.warns(UserWarning, match="ignored"):
        eng = Engine("FlowNetS", caffe, 1, 64, 64, "f32")
    assert sorted(eng.ignored_variables) == ["FlowNetS/deconv5/biases", "FlowNetS/upsample_flow6to5/biases"]
    assert torch.equal(eng(a, b)["flow"], base)


def test_interp_with_deconv_biases():
    """FlowNetS_interp(no_deconv_biases=False): predict_flowN and deconvN carry biases, upsample_flowXtoY none
    (flownet_s_interp.py:78-126)."""
    from src import weights as W
    from src.engine import Engine
    wts = W.init_weights("FlowNetS_interp", 5, head_biases=True)
    rng = np.random.default_rng(2)
    a = rng.random((1, 64, 64, 3), dtype=np.float32)
    m = (rng.random((1, 64, 64, 1)) < 0.1).astype(np.float32)
    sf = (rng.standard_normal((1, 64, 64, 2)) * 3).astype(np.float32) * m
    eng = Engine("FlowNetS_interp", wts, 1, 64, 64, "f32", no_deconv_biases=False)
    eng.set_inputs_interp(a, m, sf)
    eng.launch()
    got = eng.outputs["flow"].float().cpu().numpy()
    want = refm.flownet_s_interp(wts, {"input_a": a, "matches_a": m, "sparse_flow": sf}, no_deconv_biases=False)["flow"]
    assert epe(got, want) < EPE_TOL
    nob = {k: (np.zeros_like(v) if "/deconv" in k and k.endswith("/biases") else v) for k, v in wts.items()}
    assert epe(refm.flownet_s_interp(nob, {"input_a": a, "matches_a": m, "sparse_flow": sf}, no_deconv_biases=False)["flow"],
               want) > 3 * EPE_TOL   # the deconv biases matter in this test


def test_split_fp16_full_size_flownet_c():
    out, want = run("FlowNetC", "f16x2", 1, 384, 512)
    e = epe(out["flow"], want["flow"])
    print("FlowNetC 512x384 f16x2 mean EPE vs oracle %.3e px" % e)
    assert e < EPE_TOL


@pytest.mark.parametrize("model", ["FlowNetS", "FlowNetC", "FlowNet2"])
def test_f16_path_reported(model):
    """fp16 activations/weights (11 mantissa bits), fp32 accumulate: measured against the oracle and
    printed; asserted only against a loose sanity bound -- the 1e-3 px bar is the f32 path's."""
    n, h, w = (1, 64, 64) if model == "FlowNet2" else (2, 64, 128)
    out, want = run(model, "f16", n, h, w)
    e = epe(out["flow"], want["flow"])
    mag = float(np.sqrt((want["flow"] ** 2).sum(-1)).mean())
    print(model, "f16 mean EPE vs oracle %.3e px (mean |flow| %.2f px)" % (e, mag))
    assert e < 0.01 * max(mag, 0.1)


def test_graph_replay_equals_eager():
    from src import weights as W
    from src.engine import Engine
    wts = W.init_weights("FlowNetC", 1234)
    a, b = images(1, 64, 64, 3)
    eng = Engine("FlowNetC", wts, 1, 64, 64, "f32")
    eager = eng(a, b)["flow"].clone()
    eng.capture()
    a2, b2 = images(1, 64, 64, 4)
    eng(a2, b2)
    torch.cuda.synchronize()
    replay = eng(a, b)["flow"].clone()
    torch.cuda.synchronize()
    assert torch.equal(eager, replay)


@pytest.mark.gpu
def test_test_batch_writes_outputs_and_sintel_metrics(tmp_path, golden_dir):
    """Net.test_batch (net.py:632-1000): list file of `img1 img2 gt`, batched through the engine; per-frame
    flows equal the single-pair path, the metrics log holds one block per frame + the averages."""
    from src import flowlib
    from src.flownet_s.flownet_s import FlowNetS
    from src.net import Mode
    s = os.path.join(golden_dir, "samples")
    seq = tmp_path / "alley_1"
    seq.mkdir()
    import shutil
    for i, name in enumerate(("frame_0001", "frame_0002", "frame_0003")):
        shutil.copy(os.path.join(s, "0img%d.ppm" % (i % 2)), seq / (name + ".ppm"))
    lst = tmp_path / "sintel_clean.txt"
    gt = os.path.join(s, "0flow.flo")
    lst.write_text("%s %s %s\n%s %s %s\n" % (seq / "frame_0001.ppm", seq / "frame_0002.ppm", gt,
                                             seq / "frame_0002.ppm", seq / "frame_0003.ppm", gt))
    net = FlowNetS(mode=Mode.TEST, dtype="f16x2")
    out = tmp_path / "out"
    flows = net.test_batch(None, str(lst), str(out), accumulate_metrics=True, batch_size=2)
    assert len(flows) == 2 and flows[0].shape == (384, 512, 2)
    single = net.test(None, str(seq / "frame_0001.ppm"), str(seq / "frame_0002.ppm"), out_path=str(tmp_path / "one"),
                      save_image=False, save_flo=False)
    assert np.abs(flows[0] - single).max() < 1e-4
    for name in ("frame_0001", "frame_0002"):
        assert np.array_equal(flowlib.read_flow(str(out / "alley_1" / (name + "_flow.flo"))),
                              flows[int(name[-1]) - 1])
        assert (out / "alley_1" / (name + "_viz.png")).exists()
        assert (out / "alley_1" / (name + "_viz_norm_gt_max_motion.png")).exists()
    log = (out / "sintel_clean_metrics.log").read_text()
    assert log.count("MPI-Sintel Flow Error Metrics") == 3 and "(AVERAGE)" in log and "frame_0002" in log
    m, *_ = flowlib.compute_all_metrics(flows[0], flowlib.read_flow(gt))
    assert ("%.4f" % m["EPEall"]) in log


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "f16x2"])
def test_flownet_s_interp_matches_oracle(dtype):
    """FlowNetS_interp (flownet_s_interp.py:21-156): image + sparse flow + match mask -> dense flow."""
    from src import weights as W
    from src.flownet_s_interp.flownet_s_interp import FlowNetS_interp
    from src.net import Mode
    rng = np.random.default_rng(9)
    H, Wd = 128, 192  # predict_flow6 must be at least 2x2 for the loss downsample
    a = rng.random((2, H, Wd, 3), dtype=np.float32)
    m = (rng.random((2, H, Wd, 1)) > 0.9).astype(np.float32)
    sf = (rng.standard_normal((2, H, Wd, 2)) * 8).astype(np.float32) * m
    net = FlowNetS_interp(mode=Mode.TEST, dtype=dtype)
    wts = net.load_weights(None, seed=21)
    inputs = {"input_a": a, "matches_a": m, "sparse_flow": sf}
    got = net.model(inputs)
    want = refm.flownet_s_interp(wts, inputs)
    for k in ("predict_flow6", "predict_flow2", "flow"):
        g = got[k].float().cpu().numpy()
        assert refm.average_endpoint_error(want[k], g) / (g.shape[1] * g.shape[2]) < 1e-3
    assert set(net.model(inputs, is_training=False)) == {"flow"}
    # loss: five-scale HFEM ('hard') + L2 against the NumPy restatement
    gt = (rng.standard_normal((2, H, Wd, 2)) * 5).astype(np.float32)
    total, aepe = net.loss(gt, got, add_hard_flow_mining="hard", lambda_weight=2.0, hard_examples_perc=50)
    from oracle import ops as refops
    t = gt * np.float32(0.05)
    ref = 0.0
    for lvl, wgt in zip((6, 5, 4, 3, 2), (0.32, 0.08, 0.02, 0.01, 0.005)):
        p = want["predict_flow%d" % lvl]
        ref += wgt * refm.average_endpoint_error_hfem(refops.downsample(t, p.shape[1:3]), p, "hard", 2.0, 50)
    ref = ref / 5.0 + sum(0.5 * 4e-4 * float(np.sum(np.square(v.astype(np.float64)))) for k, v in wts.items()
                          if k.endswith("/weights") and "deconv" not in k and "upsample_flow" not in k)
    assert float(total) == pytest.approx(ref, rel=2e-4)
    assert float(aepe) == pytest.approx(refm.mean_endpoint_error(t, want["flow"]), rel=1e-4)


@pytest.mark.gpu
def test_interp_cli_path_writes_flow(tmp_path, golden_dir):
    from PIL import Image
    from src import flowlib
    from src.flownet_s_interp.flownet_s_interp import FlowNetS_interp
    from src.net import Mode
    s = os.path.join(golden_dir, "samples")
    gt = flowlib.read_flow(os.path.join(s, "0flow.flo"))
    mask = (np.random.default_rng(2).random(gt.shape[:2]) > 0.95)
    Image.fromarray((mask * 255).astype(np.uint8)).save(tmp_path / "mask.png")
    flowlib.write_flow(gt * mask[..., None], str(tmp_path / "sparse.flo"))
    net = FlowNetS_interp(mode=Mode.TEST, dtype="f16x2")
    flow = net.test(None, os.path.join(s, "0img0.ppm"), matches_a_path=str(tmp_path / "mask.png"),
                    sparse_flow_path=str(tmp_path / "sparse.flo"), input_type="image_matches",
                    out_path=str(tmp_path), gt_flow=os.path.join(s, "0flow.flo"))
    assert flow.shape == (384, 512, 2) and np.isfinite(flow).all()
    assert np.array_equal(flowlib.read_flow(str(tmp_path / "samples" / "0img0_flow.flo")), flow)
    with pytest.raises(ValueError):
        net.test(None, os.path.join(s, "0img0.ppm"), input_type="image_matches", out_path=str(tmp_path))


@pytest.mark.gpu
def test_sintel_shape_through_adapt_x():
    """BASELINE config 5 shape: a 436 x 1024 pair is zero-padded to 448 x 1024 (net.py:373-388), run, cropped."""
    from src.flownet_c.flownet_c import FlowNetC
    from src.net import Mode
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, (436, 1024, 3)).astype(np.uint8)
    b = np.roll(a, (2, -4), (0, 1))
    net = FlowNetC(mode=Mode.TEST, dtype="f16x2")
    wts = net.load_weights(None, seed=5)
    a1, b1, info = net.adapt_x(a, b)
    assert a1.shape == (1, 448, 1024, 3) and info == (1, 436, 1024, 3)
    got = net.model({"input_a": a1, "input_b": b1})["flow"][0].float().cpu().numpy()
    got = net.postproc_y_hat_test(got, (info[-3], info[-2], 2))
    want = refm.flownet_c(wts, {"input_a": a1, "input_b": b1})["flow"][0, :436]
    assert got.shape == (436, 1024, 2)
    d = got.astype(np.float64) - want
    assert float(np.sqrt((d * d).sum(-1)).mean()) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "f16x2"])
def test_sample_pair_against_the_committed_fixture(golden_dir, dtype):
    """The engine on the reference's sample pair against tests/golden/flownets_sample0_golden.npz (no oracle run)."""
    import sys
    sys.path.insert(0, golden_dir)
    import make_golden_flownets as gen
    from src import weights as W
    from src.engine import Engine
    g = np.load(os.path.join(golden_dir, "flownets_sample0_golden.npz"))
    a, b = gen.inputs()
    out = Engine("FlowNetS", W.init_weights("FlowNetS", 1234), 1, 384, 512, dtype)(a, b)
    pf6 = out["predict_flow6"].float().cpu().numpy()
    flow = out["flow"].float().cpu().numpy()
    assert np.sqrt(((pf6 - g["predict_flow6"]) ** 2).sum(-1)).mean() < 1e-3
    assert np.sqrt(((flow[0, g["probe_y"], g["probe_x"]] - g["flow_probes"]) ** 2).sum(-1)).mean() < 1e-3


@pytest.mark.gpu
def test_uint8_input_path_is_byte_identical_to_the_fp32_path():
    """Net.adapt_x (net.py:338-392) divides uint8 frames by 255 on the host and ships float32; the uint8 path ships the
    bytes and divides on the device through a table holding the host arithmetic.  Same flow, bit for bit -- through
    the engine, through Net.model, graph replay included; an image whose max is <= 1 is left undivided as adapt_x does."""
    import ctypes as C
    from src import _hip, weights as W
    from src.engine import Engine
    from src.flownet_c.flownet_c import FlowNetC
    from src.net import Mode
    rng = np.random.default_rng(21)
    a8 = rng.integers(0, 256, (436, 500, 3)).astype(np.uint8)
    b8 = np.roll(a8, (2, -3), (0, 1))
    net = FlowNetC(mode=Mode.TEST, dtype="f16x2")
    wts = net.load_weights(None, seed=5)
    af, bf, info = net.adapt_x(a8, b8)
    au, bu, info_u, scale = net.adapt_x_u8(a8, b8)
    assert info == info_u == (1, 436, 500, 3) and scale == (True, True) and au.shape == (1, 448, 512, 3) and au.dtype == np.uint8
    want = net.model({"input_a": af, "input_b": bf})["flow"]
    got = net.model({"input_a": au, "input_b": bu, "scale": scale})["flow"]
    assert torch.equal(got, want)
    eng = net.engine(1, 448, 512, uint8_inputs=True)
    assert torch.equal(eng.in_a, torch.from_numpy(af).cuda()) and torch.equal(eng.in_b, torch.from_numpy(bf).cuda())
    eng.capture()
    eng.set_inputs_u8(bu, au)            # other inputs through the captured plan ...
    eng.launch()
    eng.set_inputs_u8(au, bu)            # ... and back
    eng.launch()
    torch.cuda.synchronize()
    assert torch.equal(eng.outputs["flow"], want)
    # an image of zeros and ones: max <= 1, adapt_x leaves it as it is (no division)
    ones = (rng.random((64, 64, 3)) < 0.5).astype(np.uint8)
    f1, f2, _ = net.adapt_x(ones, ones)
    u1, u2, _, sc = net.adapt_x_u8(ones, ones)
    assert sc == (False, False) and f1.max() == 1.0
    assert torch.equal(net.model({"input_a": u1, "input_b": u2, "scale": sc})["flow"],
                       net.model({"input_a": f1, "input_b": f2})["flow"])
    # the C entry point on a ragged count
    lib = _hip.lib()
    src = torch.from_numpy(rng.integers(0, 256, 1000003).astype(np.uint8)).cuda()
    lut = torch.from_numpy((np.arange(256) / 255.0).astype(np.float32)).cuda()
    dst = torch.full((1000003,), -1.0, device="cuda")
    _hip.check(lib.fn2_u8_to_f32_lut(_hip.ptr(src), _hip.ptr(lut), _hip.ptr(dst), 1000003, _hip.stream_ptr()))
    assert np.array_equal(dst.cpu().numpy(), (src.cpu().numpy() / 255.0).astype(np.float32))
    with pytest.raises(ValueError):
        Engine("FlowNetC", wts, 1, 64, 64, "f32").set_inputs_u8(u1, u2)


@pytest.mark.gpu
def test_model_runs_a_batch_above_the_tensor_limit_in_chunks(monkeypatch):
    """The LDS-DMA kernels address one tensor through 31-bit offsets, so a batch whose largest activation tensor would
    reach 2 GiB cannot be one engine (limits).  Net.model then runs the batch as chunks of the largest
    batch that fits: same predictions, in the caller's order.  Here the limit is lowered so that 5 pairs of 128 x 192 do
    not fit (conv1: 5 x 64 x 96 x 64 channels x 4 bytes = 7.9 MB against a limit of 4 MiB -> 2 pairs per engine)."""
    from src.engine import BatchTooLarge, Engine
    from src.flownet_s.flownet_s import FlowNetS
    from src.net import Mode
    rng = np.random.default_rng(8)
    a = rng.random((5, 128, 192, 3), dtype=np.float32)
    b = np.roll(a, (1, -2), (1, 2))
    net = FlowNetS(mode=Mode.TEST, dtype="f16x2")
    net.load_weights(None, seed=3)
    whole = net.model({"input_a": a, "input_b": b})
    monkeypatch.setattr(Engine, "MAX_TENSOR_BYTES", 4 << 20)
    with pytest.raises(BatchTooLarge) as e:
        Engine("FlowNetS", net.weights, 5, 128, 192, "f16x2")
    assert 1 <= e.value.fit < 5
    net2 = FlowNetS(mode=Mode.TEST, dtype="f16x2")
    net2.load_weights(None, seed=3)
    parts = net2.model({"input_a": a, "input_b": b})
    chunk = net2._chunk_of[(128, 192, "f16x2")]
    assert 1 <= chunk < 5 and all(k[0] <= chunk for k in net2._engines)
    assert set(parts) == set(whole)
    for k in whole:
        assert parts[k].shape == whole[k].shape
        # another batch size may pick another tile / split for a layer: equal to rounding, not bit for bit
        assert float((parts[k] - whole[k]).abs().max()) <= 2e-5 * max(1.0, float(whole[k].abs().max())), k
    # one pair that does not fit by itself is an error, not an endless retry
    monkeypatch.setattr(Engine, "MAX_TENSOR_BYTES", 1 << 20)
    net3 = FlowNetS(mode=Mode.TEST, dtype="f16x2")
    net3.load_weights(None, seed=3)
    with pytest.raises(ValueError):
        net3.model({"input_a": a[:2], "input_b": b[:2]})


@pytest.mark.gpu
@pytest.mark.parametrize("model,shape", [("FlowNetS", (2, 128, 192)), ("FlowNetSD", (1, 64, 128)), ("FlowNetS", (6, 256, 384)),
                                         ("FlowNet2", (1, 128, 256))])
def test_upsample_flow_riding_on_the_transposed_conv(model, shape, monkeypatch):
    """fn2_conv_desc.up_src: without a head lane the engine lets upsample_flow(N+1)toN (flownet_s.py:60-63) ride on the
    launch of deconvN -- in its split-K finalize pass, or as the stand-alone kernel behind a launch that has none.  Same
    arithmetic, tap for tap: every prediction equals the plan with the upsample as its own launch bit for bit, and the
    plan is shorter by the launches that rode.  The plain flow heads in front of those transposed convs ride as well
    (fn2_conv_desc.head: extra blocks of the split-K launch, or fn2_conv2d(head) in front of any other launch).  FlowNet2:
    the fusion net's fuse_upsample_flow1to0 rides in the epilogue of the merged-phase fuse_deconv0 (kind 5)."""
    from src import weights as W
    from src.engine import Engine
    n, h, w = shape
    wts = W.init_weights(model, 77)
    a, b = images(n, h, w, 9)
    monkeypatch.setenv("FN2_BRANCHES", "0")   # no head lane: the FlowNet2 stack's setting for its sub-networks
    monkeypatch.setenv("FN2_UP_IN_DECONV", "0")
    base = Engine(model, wts, n, h, w, "f16x2")
    want = {k: v.clone() for k, v in base(a, b).items()}
    monkeypatch.setenv("FN2_UP_IN_DECONV", "1")
    eng = Engine(model, wts, n, h, w, "f16x2")
    got = eng(a, b)
    rode = [d for d in eng.conv_descs if d.up_src]
    heads = [d for d in eng.conv_descs if d.head]     # plain flow heads taken along by the transposed conv behind them
    assert rode and heads and len(eng.ops) == len(base.ops) - len(rode) - len(heads)
    assert any(bool(eng.lib.fn2_conv2d_workspace_bytes(d)) for d in rode)   # finalize passes did it
    for k in want:
        assert torch.equal(got[k], want[k]), k
    # a launch without split-K (here: the workspace taken away) is followed by the stand-alone kernel: same flow slice
    import ctypes as C
    from src import _hip
    rec = [r for r in eng.layers if r["kind"] == "upflow" and any(d.up_c0 == r["dst"][1] and d.out.data == r["dst"][0].data_ptr()
                                                                   for d in rode)][0]
    d = [d for d in rode if d.up_c0 == rec["dst"][1] and d.out.data == rec["dst"][0].data_ptr()][0]
    buf, c0, _ = rec["dst"]
    before = buf.clone()
    raw = buf.view(torch.float32) if buf.dtype != torch.float32 else buf
    g0 = c0 // 8 * 8
    raw[..., g0:g0 + 8] = 0          # the 8-channel group holding the two flow channels (split fp16: hi / lo halves)
    ws, wsb = d.workspace, d.workspace_bytes
    d.workspace, d.workspace_bytes = None, 0
    _hip.check(eng.lib.fn2_conv2d(C.byref(d), _hip.stream_ptr()))
    d.workspace, d.workspace_bytes = ws, wsb
    torch.cuda.synchronize()
    assert torch.equal(buf.view(torch.float32)[..., g0:g0 + 8], before.view(torch.float32)[..., g0:g0 + 8])
    eng.capture()
    eng(a, b)
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(eng.outputs[k], want[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("model,shape", [("FlowNetS", (2, 256, 384)), ("FlowNetSD", (2, 128, 192))])
def test_head_gemm_slabs_summed_by_the_tail(model, shape, monkeypatch):
    """GEMM-form flow heads whose 1x1 GEMM splits K keep the raw slabs (fn2_conv_desc.raw_partials) and the tail launch sums
    them: the plan loses one finalize launch per such head and every prediction stays bit for bit what the finalize form
    gives (same additions in the same order)."""
    from src import weights as W
    from src.engine import Engine
    n, h, w = shape
    wts = W.init_weights(model, 78)
    a, b = images(n, h, w, 10)
    monkeypatch.setenv("FN2_HEAD_GEMM_MIN", "2048")   # predict_flow3 of this size as a GEMM too: its K (386 channels) splits
    monkeypatch.setenv("FN2_HEAD_SLABS", "0")
    base = Engine(model, wts, n, h, w, "f16x2")
    want = {k: v.clone() for k, v in base(a, b).items()}
    monkeypatch.setenv("FN2_HEAD_SLABS", "1")
    eng = Engine(model, wts, n, h, w, "f16x2")
    raw = [d for d in eng.conv_descs if d.raw_partials]
    assert raw and all(eng.lib.fn2_conv2d_splits(d) > 1 for d in raw)
    got = eng(a, b)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    eng.capture()
    eng(a, b)
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(eng.outputs[k], want[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("model,shape", [("FlowNetCSS", (2, 128, 192)), ("FlowNet2", (1, 128, 256))])
def test_flow_resize_inside_the_consuming_op(model, shape, monkeypatch):
    """Inside a stack the flow of a sub-network -- resize_bilinear(scale * predict_flow2), flownet_s.py:105-109 -- is
    interpolated per pixel by the op that consumes it (fn2_stack_input_pf: warp + brightness error + concat,
    flownet_cs.py:21-36; fn2_fusion_input_pf: flownet2.py:25-47), which also writes the flow buffer: one launch less per
    sub-network on the chain, every tensor bit for bit what resize + op give."""
    from src import weights as W
    from src.engine import Engine
    n, h, w = shape
    wts = W.init_weights(model, 79)
    a, b = images(n, h, w, 11)
    monkeypatch.setenv("FN2_FLOW_IN_CONSUMER", "0")
    base = Engine(model, wts, n, h, w, "f16x2")
    want = {k: v.clone() for k, v in base(a, b).items()}
    monkeypatch.setenv("FN2_FLOW_IN_CONSUMER", "1")
    eng = Engine(model, wts, n, h, w, "f16x2")
    got = eng(a, b)
    nsub = 2 if model == "FlowNetCSS" else 4
    assert len(eng.ops) == len(base.ops) - nsub
    for k in want:
        assert torch.equal(got[k], want[k]), k
    flows = [k for k in base.bufs if k.endswith("/flow")]
    assert len(flows) >= nsub
    for k in flows:     # the sub-networks' full-resolution flows are still there, written by their consumers
        assert torch.equal(eng.bufs[k], base.bufs[k]), k
