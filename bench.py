#!/usr/bin/env python
"""Benchmark of the hot path: FlowNet forward on synthetic 512x384 pairs (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W [--model FlowNet2 --batch 4 --dtype f16x2|f32|bf16]

Default workload = BASELINE config 3, the north-star target: FlowNet2 full stack, batch 4, 512x384, one MI355X.
A step = one forward pass of one batch of image pairs already resident in HBM.  The timed region (exactly K steps
between barrier + synchronize) is repeated `--regions` times; `value` / `ms_per_step` are the MEDIAN region and the
line carries min / max.  BASELINE config 2 (FlowNetC batch 8) is measured in the same process and reported under
`extra`.  N > 1: one rank per GPU under torch.distributed.run -- started by the driver, or by this script itself when
it is invoked with --gpus N and no RANK in the environment (child processes, before any GPU call in the parent).
Pairs are independent, so ranks shard the batch with no data-path collective (weak scaling: `batch` pairs per GPU).
Rank 0 prints ONE JSON line (contract in the task statement; extra keys documented in DESIGN.md).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "flownet2-tf_amd"))
sys.path.insert(0, ROOT)

# f16x2: 3 fp16 MFMAs per algorithmic product -> the algorithmic peak is a third of the fp16 peak
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0, "f16x2": 2500.0 / 3}  # dense MFMA peaks, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
K80_MS = {"FlowNetS": 38.067, "FlowNetC": 78.789, "FlowNetCS": 123.300, "FlowNetCSS": 161.186,
          "FlowNetSD": 62.061, "FlowNet2": 276.641}  # reference README.md:71 (other hardware)


def synth_pairs(n, h, w, seed0):
    """uint8 noise image; second image = first rolled by (3,-5) px + noise +-4."""
    a_l, b_l = [], []
    for i in range(n):
        rng = np.random.default_rng(seed0 + i)
        a = rng.integers(0, 256, (h, w, 3)).astype(np.float32)
        b = np.clip(np.roll(a, (3, -5), (0, 1)) + rng.uniform(-4, 4, a.shape), 0, 255)
        a_l.append(a / 255.0)
        b_l.append(b.astype(np.float32) / 255.0)
    return np.stack(a_l).astype(np.float32), np.stack(b_l).astype(np.float32)


def pad64(x):
    """Net.adapt_x (net.py:373-388): zero-pad bottom / right to multiples of 64 (1024x436 -> 1024x448)."""
    n, h, w, c = x.shape
    return np.pad(x, [(0, 0), (0, -h % 64), (0, -w % 64), (0, 0)])


def synth_gt(n, h, w, rank=0):
    """Ground-truth flow of the train workload: N(0, 5^2) px clipped to +-40."""
    rng = np.random.default_rng(77 + rank)
    return np.clip(rng.standard_normal((n, h, w, 2)) * 5, -40, 40).astype(np.float32)


def stored_traffic(model, batch, dtype, kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (tools/pmc_traffic.py: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes).  PMC collection
    cannot run inside this process, so the number is read from profiles/ and is null when no pass
    exists for this workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_%s_b%d_%s.json" % (model, batch, dtype))
    if not os.path.exists(path):
        return None
    table = json.load(open(path))
    stem = kernel.rstrip(">")  # the bench names an instantiation without its defaulted trailing template arguments
    for k, v in table.items():
        if stem in k:
            return round(v["hbm_bytes_per_launch"])
    return None


def per_kernel_times(eng, steps):
    """Eager pass with an event pair around every launch on the launch stream."""
    n_ops = len(eng.ops)
    acc = np.zeros(n_ops)
    s = None
    from src import _hip
    for _ in range(steps):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_ops + 1)]
        s = _hip.stream_ptr()
        evs[0].record()
        for i, (name, fn, args) in enumerate(eng.ops):
            rc = fn(*args, s)
            if rc:
                _hip.check(rc)
            evs[i + 1].record()
        torch.cuda.synchronize()
        for i in range(n_ops):
            acc[i] += evs[i].elapsed_time(evs[i + 1])
    return acc / steps  # ms per launch


def cpu_baseline(model, h, w, seed):
    """The NumPy oracle (CPU restatement of the reference graph) timed on the host cores:
    one pair, one run (bounded sample of the same workload)."""
    from oracle import models as refm
    from src import weights as W
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    wts = W.init_weights(model, 1234)
    fn = refm.MODELS[model]
    pairs, total = 0, 0.0
    while total < 10.0 and pairs < 16:   # bounded sample: >= ~10 s of CPU work, at most 16 pairs
        a, b = synth_pairs(1, h, w, seed + pairs)
        t0 = time.perf_counter()
        fn(wts, {"input_a": a, "input_b": b})
        total += time.perf_counter() - t0
        pairs += 1
    return {"value": pairs / total, "unit": "pairs/s", "cores": int(threads), "kind": "port",
            "sample": "%d pairs %s forward %dx%d one at a time, NumPy fp64 oracle (%.1f s)" % (pairs, model, w, h, total)}


def cpu_baseline_train(h, w):
    """The float64 autograd oracle of the train step (loss + all gradients of FlowNetS, one pair at a time) timed on the
    host cores: bounded sample (>= ~10 s, at most 8 pairs).  Adam is elementwise and not included."""
    from oracle import train as reft
    from src import weights as W
    wts = W.init_weights("FlowNetS", 1234)
    pairs, total = 0, 0.0
    while total < 10.0 and pairs < 8:
        a, b = synth_pairs(1, h, w, pairs)
        gt = synth_gt(1, h, w, pairs)
        t0 = time.perf_counter()
        reft.flownet_s_loss_and_grads(wts, a, b, gt)
        total += time.perf_counter() - t0
        pairs += 1
    return {"value": pairs / total, "unit": "pairs/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "%d pairs FlowNetS loss + gradients %dx%d one at a time, torch float64 autograd oracle (%.1f s)"
                      % (pairs, w, h, total)}


def run_train(args, rank, world, dist, full=True, batch=None):
    """BASELINE config 4: FlowNetS train step on FlyingChairs-shaped synthetic pairs, `batch` pairs per GPU,
    data parallel: the flat gradient arena is all-reduced (RCCL) in 4 buckets under the backward pass, then Adam."""
    from src import _hip, weights as W
    from src.trainer import FlowNetSTrainer
    batch = batch or args.batch
    steps = args.steps
    tr = FlowNetSTrainer(W.init_weights("FlowNetS", 1234), batch, args.height, args.width, dtype=args.train_dtype)
    peak = PEAK_TFLOPS[args.train_dtype]
    a, b = synth_pairs(batch, args.height, args.width, seed0=1000 * rank)
    gt = synth_gt(batch, args.height, args.width, rank)
    a, b, gt = (torch.as_tensor(x).cuda() for x in (a, b, gt))  # resident in HBM before the timed region

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.train_step(a, b, gt)
    tr.wait_events = []          # (before, after) event pairs around the bucket waits of every step from here on
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.train_step(a, b, gt)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss.item())
    if not np.isfinite(loss_val):
        raise RuntimeError("train step produced a non-finite loss")
    exposed = [e0.elapsed_time(e1) for e0, e1 in tr.wait_events]
    tr.wait_events = None
    # ---- the gradient exchange alone: the same buckets, nothing to hide behind (every rank takes part)
    comm = None
    if dist is not None:
        from src.dist import allreduce_bucket_async
        reps = 5
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            hs = [allreduce_bucket_async(bk) for _, bk in tr.buckets]
            for h in hs:
                if h is not None:
                    h.wait()
        e1.record()
        barrier()
        alone = e0.elapsed_time(e1) / reps
        nbytes = tr.grad_arena.numel() * 4
        exp_ms = float(np.mean(exposed)) if exposed else None
        comm = {"allreduce_ms": round(alone, 4), "allreduce_exposed_ms": None if exp_ms is None else round(exp_ms, 4),
                "overlap_frac": None if exp_ms is None or alone <= 0 else round(max(0.0, 1.0 - exp_ms / alone), 4),
                "payload_MB": round(nbytes / 1e6, 1), "buckets": len(tr.buckets),
                "busbw_GBs": round(2.0 * (world - 1) / world * nbytes / (alone * 1e-3) / 1e9, 1) if alone > 0 else None}
    if rank != 0:
        return None
    ms_step = dt / steps * 1e3
    # ---- per-launch event timing of forward + backward on the launch stream
    flops = dict(tr.eng.layer_flops)
    launches = [(n, fn, ar, k, flops.get(n, 0.0)) for (n, fn, ar), k in zip(tr.eng.ops, tr.eng.kernel_of)]
    launches += tr.backward_launches()
    acc = np.zeros(len(launches))
    reps = max(1, min(steps, 10))
    for _ in range(reps):
        tr.forward_backward(a, b, gt)  # leaves valid gradients for every backward launch to re-run on
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(launches) + 1)]
        s = _hip.stream_ptr()
        evs[0].record()
        for i, (_, fn, ar, _, _) in enumerate(launches):
            _hip.check(fn(*ar, s))
            evs[i + 1].record()
        torch.cuda.synchronize()
        for i in range(len(launches)):
            acc[i] += evs[i].elapsed_time(evs[i + 1]) / reps
    fams = {}
    for (name, _, _, kern, fl), ms in zip(launches, acc):
        d = fams.setdefault(kern, {"ms": 0.0, "launches": 0, "flop": 0.0})
        d["ms"] += ms
        d["launches"] += 1
        d["flop"] += fl
        if args.per_layer:
            sys.stderr.write("%-52s %-40s %9.4f ms %8.1f TFLOP/s\n" % (name, kern[:40], ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0))
    dom = max(fams, key=lambda k: fams[k]["ms"])
    D = fams[dom]
    avg_ms = D["ms"] / D["launches"]
    achieved = (D["flop"] / D["launches"]) / (avg_ms * 1e-3) / 1e12
    total_flop = sum(v["flop"] for v in fams.values())
    kpeak = PEAK_TFLOPS["f32"] if dom == "bwd_filter_kernel" else peak
    out = {
        "metric": "train pairs/sec at 512x384 (FlowNetS fwd+bwd+Adam)", "value": round(world * batch * steps / dt, 2),
        "unit": "pairs/s", "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.train_dtype == "f32" else "f16x2 operands (3 fp16 MFMAs per product), fp32 accumulate / masters / Adam",
        "data": "synthetic",
        "config": {"workload": "FlowNetS train step (fwd + multiscale EPE + bwd + Adam), batch=%d synthetic %dx%d pairs "
                               "per GPU, seeded synthetic weights" % (batch, args.width, args.height),
                   "pairs_per_gpu": batch,
                   "parallelism": "dp%d (the %.0f MB gradient arena all-reduced in %d buckets under the backward pass)"
                                  % (world, tr.grad_arena.numel() * 4 / 1e6, len(tr.buckets))},
        "n_ranks_seen": dist.get_world_size() if dist is not None else 1,
        "backend": (dist.get_backend() if dist is not None else None),
        "loss": round(loss_val, 6), "output_finite": True,
        "comm": comm,
        "roofline": {"kernel": dom, "bound": kernel_bound(dom), "achieved": round(achieved, 2),
                     "peak": kpeak, "unit": "TFLOP/s", "frac": round(achieved / kpeak, 4),
                     "traffic": stored_traffic("FlowNetS_train", batch, args.train_dtype, dom)
                     if (args.height, args.width) == (384, 512) else None,
                     "launches_per_step": D["launches"], "avg_launch_ms": round(avg_ms, 5),
                     "flop_per_launch": D["flop"] / D["launches"]},
        "kernels": {k: {"ms_per_step": round(v["ms"], 4), "launches": v["launches"]} for k, v in fams.items()},
        "model_tflops": round(total_flop / (ms_step * 1e-3) / 1e12, 2),
        "step": {"mfma_frac": round(total_flop / (ms_step * 1e-3) / 1e12 / peak, 4)},
    }
    if not full:
        for k in ("steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "data", "n_gpus", "kernels"):
            out.pop(k, None)
        return out
    out["cpu_baseline"] = None if (args.no_cpu_baseline or world != 1) else cpu_baseline_train(args.height, args.width)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--regions", type=int, default=5, help="repeats of the timed region (each exactly --steps steps); "
                                                           "the median region is reported, min / max beside it")
    ap.add_argument("--model", default="FlowNet2", help="default: BASELINE config 3 (FlowNet2 full stack, batch 4, 512x384)")
    ap.add_argument("--batch", type=int, default=None, help="pairs per GPU per step (default 4; 8 for --mode train / FlowNetC)")
    ap.add_argument("--no-extra", dest="extra", action="store_false",
                    help="skip the FlowNetC batch-8 line (BASELINE config 2) reported under `extra`")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--dtype", default="f16x2", choices=["f32", "bf16", "f16", "f16x2"],
                    help="f16x2 (default) and f32 meet the 1e-3 px parity bar; bf16/f16 do not")
    ap.add_argument("--mode", default="forward", choices=["forward", "train"],
                    help="train: FlowNetS fwd + multiscale EPE loss + bwd + Adam (+ gradient all-reduce for N > 1), fp32")
    ap.add_argument("--train-dtype", default="f16x2", choices=["f32", "f16x2"],
                    help="--mode train: f16x2 = split-fp16 activations / gradients / weight copies, fp32 masters and Adam")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--per-layer", action="store_true", help="print per-launch ms and TFLOP/s to stderr")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 8 if (args.mode == "train" or args.model == "FlowNetC") else 4

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` on its own: start N fresh rank processes under torch.distributed.run BEFORE this
        # process touches the GPU (device_count() does not initialise it), relay their output and exit with their code.
        ndev = torch.cuda.device_count()
        env = dict(os.environ)
        if ndev < args.gpus:
            # fewer devices than ranks (a one-GPU box): rehearsal of the identical control flow with ranks sharing the
            # devices round-robin and the collectives on gloo; RCCL needs one device per rank
            if args.gpus > 6 * max(ndev, 1):
                raise SystemExit("--gpus %d on %d device(s): more than 6 ranks per device" % (args.gpus, ndev))
            env.setdefault("FN2_BENCH_BACKEND", "gloo")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # FN2_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (all ranks share
    # the visible devices round-robin, collectives staged through the host); the driver's runs use nccl = RCCL.
    backend = os.environ.get("FN2_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dist = None
    # FN2_DIST_SINGLE=1 with --gpus 1: a process group of ONE rank on nccl -- the one-GPU rehearsal of the RCCL call
    # sequence of the N > 1 run (communicator, barriers, bucketed async all-reduce between the graph segments)
    if world > 1 or os.environ.get("FN2_DIST_SINGLE") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        # RCCL prints a version banner and gloo its connection lines on STDOUT while the group forms: keep stdout for the
        # one JSON line by pointing fd 1 at stderr until the first collective has run
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
                warm = torch.zeros(1, device="cuda")
                dist.all_reduce(warm)
                torch.cuda.synchronize()
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    if args.mode == "train":
        out = run_train(args, rank, world, dist)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(out))
        return

    out = forward_line(args, args.model, args.batch, rank, world, dist, full=True)
    if args.extra and (args.height, args.width) == (384, 512) and (args.model, args.batch) == ("FlowNet2", 4):
        # The other workloads BASELINE names, in the same process (every rank takes part: same barriers): configs[1]
        # FlowNetC batch 8, the metric's third model FlowNetS (batch 8) and configs[3], the FlowNetS train step with 8
        # pairs per GPU -- for N > 1 that is the one path with a collective (bucketed gradient all-reduce over RCCL)
        extra = {}
        for key, model in (("FlowNetC_b8", "FlowNetC"), ("FlowNetS_b8", "FlowNetS")):
            extra[key] = forward_line(args, model, 8, rank, world, dist, full=False)
        torch.cuda.empty_cache()
        extra["FlowNetS_train_b8"] = run_train(args, rank, world, dist, full=False, batch=8)
        if rank == 0:
            out["extra"] = extra
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def golden_epe(model, batch, h, w, dtype, flow):
    """Mean endpoint error (px) of the timed plan's flow field against the oracle probes committed for exactly this
    workload (tests/golden/plan_<model>_b<batch>_<h>x<w>.npz, rank-0 inputs); None when no fixture exists for it."""
    path = os.path.join(ROOT, "tests", "golden", "plan_%s_b%d_%dx%d.npz" % (model.lower(), batch, h, w))
    if not os.path.exists(path):
        return None
    g = np.load(path)
    f = flow.float().cpu().numpy()[:, g["probe_y"], g["probe_x"]].astype(np.float64)
    d = f - g["flow"].astype(np.float64)
    e = float(np.sqrt((d * d).sum(-1)).mean())
    if dtype in ("f32", "f16x2") and not e < 1e-3:
        raise RuntimeError("%s b%d: mean EPE vs the oracle fixture %.3e px exceeds the 1e-3 px parity bar" % (model, batch, e))
    return float("%.3e" % e)


def kernel_bound(kernel):
    """Roofline that bounds a device kernel: the implicit-GEMM convolutions and filter gradients are dense
    contractions on the matrix cores; everything else on this path is an HBM pass."""
    return "mfma" if kernel.startswith(("conv_igemm", "bwd_filter")) else "hbm"


def forward_line(args, model, batch, rank, world, dist, full):
    """Build the engine of (model, batch), time `regions` x exactly `steps` graph replays, rank 0 returns the line."""
    from src import weights as W
    from src.engine import Engine
    wts = W.init_weights(model, 1234)
    # --height 436 --width 1024 (BASELINE config 5's Sintel frames): the pairs are zero-padded to multiples of 64 as
    # Net.adapt_x does (net.py:373-388) and the engine runs at the padded size, 448 x 1024
    a, b = synth_pairs(batch, args.height, args.width, seed0=1000 * rank)
    a, b = pad64(a), pad64(b)
    eng = Engine(model, wts, batch, a.shape[1], a.shape[2], args.dtype)
    eng.set_inputs(a, b)  # inputs resident in HBM before the timed region
    torch.cuda.synchronize()
    if not args.no_graph:
        eng.capture()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.launch()
    region_s = []
    for _ in range(max(1, args.regions)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.launch()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        region_s.append(dt)
    dt = float(np.median(region_s))  # the reported region: K steps, max over ranks, median over the repeats
    ms_step = dt / args.steps * 1e3
    pairs_s = world * batch * args.steps / dt
    # ---- what was timed is what is checked: the flow field of the last replay must be finite on every rank, and on
    # rank 0 (whose inputs are the ones tests/golden/make_golden_bench_plans.py used) it is compared with the committed
    # oracle probes of this very plan -- data files, no oracle code runs here
    flow = eng.outputs["flow"]
    if not bool(torch.isfinite(flow).all().item()):
        raise RuntimeError("%s b%d: non-finite values in the flow field after the timed regions" % (model, batch))
    if rank != 0:
        return None
    fixture_epe = golden_epe(model, batch, a.shape[1], a.shape[2], args.dtype, flow) if args.height in (384, 436) else None

    # ---- per-kernel event timing on the launch stream (eager, same K steps)
    graph, eng.graph = eng.graph, None
    per_op = per_kernel_times(eng, args.steps)
    eng.graph = graph
    fams = {}
    flops = dict(eng.layer_flops)
    nbytes = dict(eng.layer_bytes)
    iobytes = dict(eng.layer_io_bytes)
    for (name, fn, _), kern, ms in zip(eng.ops, eng.kernel_of, per_op):
        d = fams.setdefault(kern, {"ms": 0.0, "launches": 0, "flop": 0.0, "bytes": 0.0, "io": 0.0})
        d["ms"] += ms
        d["launches"] += 1
        d["flop"] += flops.get(name, 0.0)
        d["bytes"] += nbytes.get(name, 0.0)
        d["io"] += iobytes.get(name, 0.0)
    if args.per_layer:
        for (name, fn, _), ms in zip(eng.ops, per_op):
            fl = flops.get(name, 0.0)
            sys.stderr.write("%-48s %9.4f ms %8.1f TFLOP/s\n" % (name, ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0))
    dom = max(fams, key=lambda k: fams[k]["ms"])
    D = fams[dom]
    avg_ms = D["ms"] / D["launches"]
    traffic = stored_traffic(model, batch, args.dtype, dom) if (args.height, args.width) == (384, 512) else None
    if kernel_bound(dom) == "mfma":
        achieved = (D["flop"] / D["launches"]) / (avg_ms * 1e-3) / 1e12
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS[args.dtype],
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_TFLOPS[args.dtype], 4), "traffic": traffic,
                    "launches_per_step": D["launches"], "avg_launch_ms": round(avg_ms, 5),
                    "flop_per_launch": D["flop"] / D["launches"],
                    # input slice + output slice + weights of the layers this instantiation runs, once each
                    "algorithmic_bytes_per_launch": round(D["io"] / D["launches"]),
                    "traffic_over_algorithmic": None if not traffic or not D["io"] else round(traffic / (D["io"] / D["launches"]), 3)}
    else:
        gbs = (D["bytes"] / D["launches"]) / (avg_ms * 1e-3) / 1e9 if D["bytes"] else None
        roofline = {"kernel": dom, "bound": "hbm", "achieved": None if gbs is None else round(gbs, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": None if gbs is None else round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "launches_per_step": D["launches"], "avg_launch_ms": round(avg_ms, 5),
                    "bytes_per_launch": D["bytes"] / D["launches"]}
    kernels = {k: {"ms_per_step": round(v["ms"], 4), "launches": v["launches"]} for k, v in fams.items()}
    ms_regions = [r / args.steps * 1e3 for r in region_s]
    out = {
        "metric": "forward pairs/sec at 512x384 (%s)" % model, "value": round(pairs_s, 2),
        "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4), "ms_per_pair": round(ms_step / batch, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f16x2": "f16x2 (split fp16 hi+lo operands, 3 fp16 MFMAs per product, fp32 accumulate)"}.get(
            args.dtype, args.dtype),
        "data": "synthetic", "config": {"workload": "%s forward, batch=%d synthetic %dx%d pairs per GPU, "
                                        "seeded synthetic weights" % (model, batch, args.width, args.height),
                                        "pairs_per_gpu": batch, "parallelism": "dp%d (no collective)" % world},
        "regions": {"n": len(ms_regions), "ms_per_step_min": round(min(ms_regions), 4),
                    "ms_per_step_median": round(ms_step, 4), "ms_per_step_max": round(max(ms_regions), 4)},
        "n_ranks_seen": dist.get_world_size() if dist is not None else 1,
        "backend": (dist.get_backend() if dist is not None else None),
        "roofline": roofline, "kernels": kernels,
        "model_gflop_per_pair": round(eng.flops_per_forward / batch / 1e9, 3),
        "model_tflops": round(eng.flops_per_forward * args.steps / dt / 1e12, 2),
        # the step as a whole against both roofs: algorithmic FLOP over the MFMA peak of the dtype; every layer's input +
        # output + weights once (plus the HBM-bound ops' SURVEY 8d bytes) over the HBM peak
        "step": {"mfma_frac": round(eng.flops_per_forward / (ms_step * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype], 4),
                 "hbm_frac": round((sum(iobytes.values()) + sum(nbytes.values())) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "algorithmic_GB": round((sum(iobytes.values()) + sum(nbytes.values())) / 1e9, 3)},
        "output_finite": True, "epe_vs_oracle_fixture_px": fixture_epe,
        "reference_k80_ms_per_pair": K80_MS.get(model),
        "speedup_vs_reference_k80": round(K80_MS[model] / (ms_step / batch), 1) if model in K80_MS else None,
    }
    if not full:
        for k in ("steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "data", "n_gpus"):
            out.pop(k, None)
        return out
    out["host_staged"] = host_staged(args, model, batch, wts, a, b, eng)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(model, args.height, args.width, seed=0)
    else:
        out["cpu_baseline"] = None
    return out


def host_staged(args, model, batch, wts, a, b, eng):
    """PCIe-inclusive rates (never `value`): the step with both image batches copied from pinned host memory and the
    flow field copied back.  Images cross the link as uint8 (what image files decode to; the `/ 255.0` of Net.adapt_x
    runs on the device, Engine.set_inputs_u8): serialised on one stream, and as a serving loop would overlap it."""
    from src.engine import Engine
    res = {}
    try:
        eng8 = Engine(model, wts, batch, a.shape[1], a.shape[2], args.dtype, uint8_inputs=True)
        a8 = torch.from_numpy(np.round(a * 255.0).astype(np.uint8)).pin_memory()
        b8 = torch.from_numpy(np.round(b * 255.0).astype(np.uint8)).pin_memory()
        eng8.set_inputs_u8(a8, b8)
        if not args.no_graph:
            eng8.capture()
        hflow = torch.empty(tuple(eng8.outputs["flow"].shape), dtype=torch.float32).pin_memory()
        nst = max(3, min(args.steps, 10))
        for timed in (False, True):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(nst):
                eng8.set_inputs_u8(a8, b8)
                eng8.launch()
                hflow.copy_(eng8.outputs["flow"], non_blocking=True)
            torch.cuda.synchronize()
            dth = time.perf_counter() - t1
        res = {"ms_per_step": round(dth / nst * 1e3, 4), "pairs_per_s": round(batch * nst / dth, 2),
               "note": "uint8 images H2D from pinned memory + device-side /255 + flow D2H inside the step, one stream, no overlap"}
        # the same as a serving loop would run it: the next batch's H2D (copy stream, double-buffered device staging) and
        # the previous flow's D2H (second copy stream) under the current forward; the forward's stream only adds a D2D
        # copy of the staged bytes into the engine's uint8 input buffers
        main_s = torch.cuda.current_stream()
        h2d_s, d2h_s = torch.cuda.Stream(), torch.cuda.Stream()
        stage = [(torch.empty_like(eng8.in_a_u8), torch.empty_like(eng8.in_b_u8)) for _ in range(2)]
        flow_dev = [torch.empty_like(eng8.outputs["flow"]) for _ in range(2)]
        staged = [torch.cuda.Event() for _ in range(2)]
        consumed = [torch.cuda.Event() for _ in range(2)]
        done = [torch.cuda.Event() for _ in range(2)]
        fetched = [torch.cuda.Event() for _ in range(2)]

        def pipelined(n_steps):
            for ev in consumed + fetched:
                ev.record(main_s)
            with torch.cuda.stream(h2d_s):
                stage[0][0].copy_(a8, non_blocking=True); stage[0][1].copy_(b8, non_blocking=True)
                staged[0].record(h2d_s)
            for i in range(n_steps):
                cur, nxt = i & 1, (i + 1) & 1
                if i + 1 < n_steps:
                    with torch.cuda.stream(h2d_s):
                        h2d_s.wait_event(consumed[nxt])          # staging slot free again
                        stage[nxt][0].copy_(a8, non_blocking=True); stage[nxt][1].copy_(b8, non_blocking=True)
                        staged[nxt].record(h2d_s)
                main_s.wait_event(staged[cur])
                eng8.in_a_u8.copy_(stage[cur][0]); eng8.in_b_u8.copy_(stage[cur][1])
                consumed[cur].record(main_s)
                eng8.launch()
                main_s.wait_event(fetched[cur])                  # the D2H two steps ago has left this slot
                flow_dev[cur].copy_(eng8.outputs["flow"])
                done[cur].record(main_s)
                with torch.cuda.stream(d2h_s):
                    d2h_s.wait_event(done[cur])
                    hflow.copy_(flow_dev[cur], non_blocking=True)
                    fetched[cur].record(d2h_s)
            torch.cuda.synchronize()

        pipelined(3)
        t1 = time.perf_counter()
        pipelined(2 * nst)
        dtp = time.perf_counter() - t1
        res["overlapped_ms_per_step"] = round(dtp / (2 * nst) * 1e3, 4)
        res["overlapped_pairs_per_s"] = round(batch * 2 * nst / dtp, 2)
        res["overlapped_note"] = ("next batch's uint8 H2D and previous flow's D2H on copy streams under the forward "
                                  "(double-buffered staging + one D2D copy per step)")
    except Exception as e:  # the measurement is informational: never fail the bench line over it
        res["error"] = repr(e)[:200]
    return res


if __name__ == "__main__":
    main()
