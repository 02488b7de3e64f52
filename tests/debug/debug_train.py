"""Debug helper (GPU): step through the FlowNetS backward launches and check each input-gradient convolution
against torch autograd of the same layer on the CPU (float64)."""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flownet2-tf_amd")]
from src import _hip, weights as W  # noqa: E402
from src.trainer import FlowNetSTrainer, LOSS_WEIGHTS  # noqa: E402

n, h, w = 2, 128, 192
rng = np.random.default_rng(1)
a = rng.random((n, h, w, 3), dtype=np.float32)
b = np.clip(np.roll(a, (2, -3), (1, 2)) + rng.uniform(-0.02, 0.02, a.shape), 0, 1).astype(np.float32)
gt = np.clip(rng.standard_normal((n, h, w, 2)) * 5, -40, 40).astype(np.float32)
wts = W.init_weights("FlowNetS", 5)
tr = FlowNetSTrainer(wts, n, h, w)
from oracle import train as reft  # noqa: E402  (debug tool only)
AG = {}
reft.flownet_s_loss_and_grads(wts, a, b, gt, act_grads=AG)
recs = {f"{r['scope']}/{r['name']}": r for r in tr.eng.layers}

# run forward + loss grads by calling forward_backward with the backward list emptied
ops_all, tr.bwd_ops = tr.bwd_ops, []
tr.forward_backward(a, b, gt)
s = _hip.stream_ptr()
for name, ops in ops_all:
    rec = recs[name]
    short = name.split("/")[-1]
    if short in AG and rec["kind"] in (0, 2) and rec["cout"] != 2:
        dbuf, dc0, dc = rec["dst"]
        torch.cuda.synchronize()
        g = tr._gbuf(dbuf)[..., dc0:dc0 + dc].double().cpu().numpy()
        e = np.abs(g - AG[short])
        yv = dbuf[..., dc0:dc0 + dc].double().cpu().numpy()
        ov = AG[short + "/value"]
        print("forward %s: max abs diff %.2e (max %.2e), sign mismatches %d of %d" % (
            short, np.abs(yv - ov).max(), np.abs(ov).max(), int(((yv > 0) != (ov > 0)).sum()), ov.size))
        print("dL/d(%s output): rel err %.2e at %s (got %.3e want %.3e)" % (short, e.max() / np.abs(AG[short]).max(),
              np.unravel_index(e.argmax(), e.shape), g.flat[e.argmax()], AG[short].flat[e.argmax()]))
    for fn, args in ops:
        is_conv = fn is tr.lib.fn2_conv2d
        if is_conv:
            sbuf, sc0, sc = rec["src"]
            dbuf, dc0, dc = rec["dst"]
            gy = tr._gbuf(dbuf)[..., dc0:dc0 + dc].double().cpu()
            gx0 = tr._gbuf(sbuf)[..., sc0:sc0 + sc].double().cpu()
        if fn is tr.lib.fn2_leaky_bwd:
            dbuf, dc0, dc = rec["dst"]
            gpost = tr._gbuf(dbuf)[..., dc0:dc0 + dc].clone()
        if fn is tr.lib.fn2_head_bwd_data:
            sbuf, sc0, sc = rec["src"]
            hx0 = tr._gbuf(sbuf).double().cpu()
        arena0 = tr.grad_arena.clone()
        c3buf = recs["FlowNetS/conv3_1"]["dst"][0]
        c3g0 = tr._gbuf(c3buf)[..., 0:256].clone()
        _hip.check(fn(*args, s))
        torch.cuda.synchronize()
        dd = (tr._gbuf(c3buf)[..., 0:256] - c3g0)
        if float(dd.abs().max()) > 0:
            print("   ## %s op %s changed dL/d(conv3_1): max %.3e, at [0,15,1,156]: %.3e" % (
                name, getattr(fn, "__name__", fn), float(dd.abs().max()), float(dd[0, 15, 1, 156])))
        changed = (tr.grad_arena != arena0).nonzero().flatten()
        if changed.numel():
            lo, hi = int(changed.min()), int(changed.max())
            owners = [k for k, r in recs.items() for t in (r["dw"], r.get("db")) if t is not None and
                      (t.data_ptr() - tr.grad_arena.data_ptr()) // 4 <= hi and
                      (t.data_ptr() - tr.grad_arena.data_ptr()) // 4 + t.numel() > lo]
            if owners != [name]:
                print("   !! %s op %s touched arena of %s" % (name, fn.__name__ if hasattr(fn, "__name__") else fn, owners))
        if fn is tr.lib.fn2_head_bwd_data:
            hx1 = tr._gbuf(sbuf).double().cpu()
            wt = torch.tensor(np.asarray(wts[name + "/weights"], np.float64))
            x = torch.zeros((n, sc) + tuple(sbuf.shape[1:3]), dtype=torch.float64, requires_grad=True)
            yy = F.conv2d(x, wt.permute(3, 2, 0, 1), padding=1)
            yy.backward(tr._gbuf(rec["dst"][0]).double().cpu().permute(0, 3, 1, 2))
            want = x.grad.permute(0, 2, 3, 1)
            d = hx1 - hx0
            e_in = (d[..., sc0:sc0 + sc] - want).abs().max() / want.abs().max()
            d[..., sc0:sc0 + sc] = 0
            print("   head_bwd_data %-16s rel err %.2e, stray writes outside the slice: %.2e" % (short, float(e_in), float(d.abs().max())))
        if fn is tr.lib.fn2_leaky_bwd:
            y = dbuf[..., dc0:dc0 + dc]
            want = torch.where(y > 0, gpost, 0.1 * gpost)
            got = tr._gbuf(dbuf)[..., dc0:dc0 + dc]
            print("   leaky_bwd %-20s max abs diff %.2e (max %.2e)" % (short, float((got - want).abs().max()), float(want.abs().max())))
        if fn is tr.lib.fn2_bias_grad and rec["cout"] != 2:
            dbuf, dc0, dc = rec["dst"]
            want = tr._gbuf(dbuf)[..., dc0:dc0 + dc].double().sum((0, 1, 2))
            print("   bias_grad %-20s rel err %.2e" % (short, float((rec["db"].double() - want).abs().max() / want.abs().max())))
        if fn is tr.lib.fn2_conv2d_bwd_filter:
            sbuf, sc0, sc = rec["src"]
            dbuf, dc0, dc = rec["dst"]
            g = tr._gbuf(dbuf)[..., dc0:dc0 + dc].double().cpu().permute(0, 3, 1, 2)
            wt = torch.tensor(np.asarray(wts[name + "/weights"], np.float64), requires_grad=True)
            if rec["kind"] == 2:
                x = torch.cat([torch.tensor(a), torch.tensor(b)], 3).double().permute(0, 3, 1, 2)
            else:
                x = sbuf[..., sc0:sc0 + sc].double().cpu().permute(0, 3, 1, 2)
            if rec["kind"] == 1:
                yy = F.conv_transpose2d(x, wt.permute(3, 2, 0, 1), stride=2, padding=1)
            else:
                yy = F.conv2d(x, wt.permute(3, 2, 0, 1), stride=rec["stride"], padding=rec["pad"])
            yy.backward(g)
            gw = wt.grad.numpy().astype(np.float32)
            if rec["kind"] == 1:
                want = W.pack_deconv(gw, rec["tile"], rec["kstep"], rec["cin_pad"], rec["layout"])[0]
            elif rec["kind"] == 2:
                want = W.pack_stem(gw, rec["cs"], rec["cin_pad"], rec["tile"], rec["layout"])[0]
            else:
                want = W.pack_conv(gw, rec["tile"], rec["kstep"], rec["cin_pad"], rec["layout"])[0]
            got = rec["dw"].cpu().numpy()
            e = np.abs(got - want.reshape(-1))
            print("   bwd_filter %-19s rel err %.2e at %d of %d" % (short, e.max() / np.abs(want).max(), e.argmax(), e.size))
        if is_conv:
            torch.cuda.synchronize()
            gx1 = tr._gbuf(sbuf)[..., sc0:sc0 + sc].double().cpu()
            wt = torch.tensor(np.asarray(wts[name + "/weights"], np.float64))
            x = torch.zeros((n, sc) + tuple(sbuf.shape[1:3]), dtype=torch.float64, requires_grad=True)
            if rec["kind"] == 1:
                y = F.conv_transpose2d(x, wt.permute(3, 2, 0, 1), stride=2, padding=1)
            else:
                y = F.conv2d(x, wt.permute(3, 2, 0, 1), stride=rec["stride"], padding=rec["pad"])
            y.backward(gy.permute(0, 3, 1, 2))
            want = x.grad.permute(0, 2, 3, 1)
            got = gx1 - gx0
            err = (got - want).abs()
            print("%-28s kind %s k%d s%d  M=%d  rel err %.2e  at %s" % (
                name, rec["kind"], rec["k"], rec["stride"], got.shape[0] * got.shape[1] * got.shape[2],
                float(err.max() / (want.abs().max() + 1e-30)), np.unravel_index(int(err.argmax()), err.shape)))
