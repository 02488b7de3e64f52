"""python -m src.flownet_s.train --list train.txt --out ./logs [--steps N --batch 8 --checkpoint w.npz --dtype f16x2
                                 --ckpt-format npz|tf]
The reference's src/flownet_s/train.py:8-40 + Net.train (net.py:1002-1400) over the HIP trainer: FlyingChairs-style
augmentation on the GPU, multiscale EPE loss, Adam on LONG_SCHEDULE, periodic .npz checkpoints under the reference's
variable names.  Data: a list file of `image_a image_b flow.flo` triples (the reference's TFRecords are built from
exactly these).  Under torch.distributed.run every rank trains on its own shuffling of the list (seed + rank) and
the gradients are all-reduced (data parallel)."""
import argparse
import os
import sys
import time

import numpy as np

from ..dataloader import FLYING_CHAIRS_PREPROCESS, load_batches
from ..training_schedules import LONG_SCHEDULE, SCHEDULES


def unpack_weights(trainer):
    """{reference variable name: array in the reference layout}: the fp32 masters of the trained layers (packed
    layouts unpacked) over the frozen variables of the checkpoint the trainer was built from (the networks in front of
    the last one in FlowNetCS / CSS, CSS and SD in FlowNet2), optimizer slots left out."""
    from ..trainer import _index_hwio
    slots = ("/Adam", "/Adam_1")
    out = {k: np.asarray(v) for k, v in trainer.host_weights.items()
           if k.endswith(("/weights", "/biases")) and not k.endswith(slots)}
    for rec in trainer.layers:
        if rec["kind"] == "corr" or rec.get("shared_with") is not None:
            continue
        name = f"{rec['scope']}/{rec['name']}"
        flat = rec["master"].cpu().numpy().reshape(-1)
        if rec["kind"] == "upflow":
            out[name + "/weights"] = flat.reshape(4, 4, 2, 2).copy()
        else:
            out[name + "/weights"] = flat[_index_hwio(rec)].astype(np.float32)
        if rec.get("b") is not None:
            out[name + "/biases"] = rec["b"].cpu().numpy().copy()
    return out


def save_checkpoint(out_dir, step, weights, fmt="npz", stem="flownet_s"):
    """npz: <stem>-<step>.npz (flownet_s-<step>.npz / flownet_sd-<step>.npz).  tf: model.ckpt-<step>.{index,data-00000-of-00001} + the `checkpoint` state file,
    the files the reference's slim Saver leaves in its log dir (net.py:1386-1392) and that both this package and
    tf.train.Saver.restore read (variables under the reference names + global_step)."""
    from .. import weights as W
    if fmt == "tf":
        from .. import tf_checkpoint
        name = "model.ckpt-%d" % step
        tf_checkpoint.save_tf_checkpoint(os.path.join(out_dir, name), dict(weights, global_step=np.int64(step)))  # (+ slots)
        with open(os.path.join(out_dir, "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (name, name))
        return os.path.join(out_dir, name)
    path = os.path.join(out_dir, "%s-%d.npz" % (stem, step))
    W.save_npz(path, weights)
    return path


def load_full_checkpoint(path):
    """Everything a checkpoint holds, bookkeeping included ({name: array}): weights, Adam slots, global_step."""
    from .. import tf_checkpoint, weights as W
    path = str(path)
    for suffix in (".index", ".data-00000-of-00001", ".meta"):
        if path.endswith(suffix) and tf_checkpoint.is_tf_checkpoint(path[:-len(suffix)]):
            path = path[:-len(suffix)]
    if tf_checkpoint.is_tf_checkpoint(path):
        return tf_checkpoint.load_tf_checkpoint(path, float_only=False)
    return W.load_weights(path)


def main(flags):
    import torch
    from .. import weights as W
    from ..trainer import FlowNetSTrainer
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("FN2_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    model = getattr(flags, "model", "FlowNetS")
    wts = W.load_weights(flags.checkpoint) if flags.checkpoint else W.init_weights(model, flags.seed)
    pre = FLYING_CHAIRS_PREPROCESS
    h, w = (pre["crop_height"], pre["crop_width"]) if flags.augment else (flags.height, flags.width)
    sched_name = getattr(flags, "training_schedule", "long_schedule").lower()
    if sched_name not in SCHEDULES:
        raise ValueError("--training_schedule must be one of: %s" % ", ".join(sorted(SCHEDULES)))
    tr = FlowNetSTrainer(wts, flags.batch, h, w, schedule=SCHEDULES[sched_name], dtype=flags.dtype, model=model)
    # parameters of the computed policies (clr / one_cycle / exp_decr / lr_range_test; net.py:1139-1190)
    tr.train_params = {k: getattr(flags, k) for k in ("clr_min_lr", "clr_max_lr", "clr_stepsize", "clr_gamma", "clr_mode",
                                                      "one_cycle_annealing_factor", "start_lr", "end_lr")
                       if getattr(flags, k, None) is not None}
    if flags.checkpoint:  # Adam moments + global_step, when the checkpoint carries them (ours and the reference's do)
        state = load_full_checkpoint(flags.checkpoint)
        restored = tr.load_optimizer_state(state)
        if rank == 0 and restored:
            print("resumed %d optimizer slots at global step %d" % (restored, tr.step_count), flush=True)
    os.makedirs(flags.out, exist_ok=True)
    t0 = time.perf_counter()
    step0 = tr.step_count
    end = step0 + flags.steps  # --steps counts the steps of THIS run; `step` is the global step (schedule, file names)
    for step, (a, b, f) in enumerate(load_batches(flags.list, flags.batch, pre, flags.augment, seed=flags.seed + rank,
                                                  global_step=step0), step0 + 1):
        loss = tr.train_step(a, b, f)
        if step % flags.log_every == 0 or step == end:
            val = float(loss.item()) + (tr.l2_term() if flags.report_l2 else 0.0)
            if rank == 0:
                # rate of THIS run: steps since the resumed global step, not the global step
                print("global step %6d | loss %.5f | %.1f pairs/s" % (step, val, world * flags.batch * (step - step0) /
                                                                      (time.perf_counter() - t0)), flush=True)
        val_every = getattr(flags, "val_every", 0)
        validate = bool(getattr(flags, "val_list", None) and val_every and (step % val_every == 0 or step == end))
        save = step % flags.save_every == 0 or step == end
        if rank == 0 and validate:
            # validation frames come at their own size (--height/--width or the list's); the trainer's engine is built
            # at the training size (the augmentation crop when --augment): evaluate() centre-crops larger frames to it
            vb = load_batches(flags.val_list, flags.batch, pre, False, seed=0, epochs=1,
                              image_size=(flags.height, flags.width))
            print("global step %6d | validation EPE %.4f px" % (step, tr.evaluate(vb, getattr(flags, "val_batches", None))),
                  flush=True)
        if rank == 0 and save:
            save_checkpoint(flags.out, step, dict(unpack_weights(tr), **tr.optimizer_state()), flags.ckpt_format,
                            stem=model.lower().replace("net", "net_"))
        if world > 1 and (validate or save):
            # rank 0 alone validates / writes: the others wait here instead of inside the next step's first bucket
            # all-reduce (whose timeout a long validation would hit); every rank takes this branch at the same steps
            torch.distributed.barrier()
        if step >= end:
            break
    return tr


def parse_and_run(model):
    ap = argparse.ArgumentParser()
    ap.add_argument("--list", required=True, help="text file of `image_a image_b flow.flo` triples, or a .tfrecords file of the "
                                                  "reference's converter (python -m src.tfrecord builds one)")
    ap.add_argument("--out", required=True, help="directory for the .npz checkpoints")
    ap.add_argument("--checkpoint", default=None, help=".npz / .npy / TensorFlow checkpoint prefix to continue from")
    ap.add_argument("--ckpt-format", default="npz", choices=["npz", "tf"],
                    help="tf: model.ckpt-<step> TensorFlow V2 bundles (readable by the reference's Saver.restore)")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=8, help="pairs per GPU (the reference's FlyingChairs batch size)")
    ap.add_argument("--dtype", default="f16x2", choices=["f32", "f16x2"])
    ap.add_argument("--no-augment", dest="augment", action="store_false")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--log-every", type=int, default=10)
    ap.add_argument("--save-every", type=int, default=1000)
    ap.add_argument("--report-l2", action="store_true", help="add the L2 regulariser to the printed loss")
    ap.add_argument("--val-list", dest="val_list", default=None, help="validation list file or .tfrecords (no augmentation)")
    ap.add_argument("--val-every", dest="val_every", type=int, default=0, help="validate every N steps (0 = never)")
    ap.add_argument("--val-batches", dest="val_batches", type=int, default=None, help="at most this many validation batches")
    ap.add_argument("--training_schedule", default="long_schedule",
                    help="long_schedule (default), fine_schedule, short_schedule, finetune_sintel_s1..5, finetune_kitti_s1..4, "
                         "finetune_rob, clr, one_cycle, exp_decr, lr_range_test (src/training_schedules.py)")
    for name, typ in (("clr_min_lr", float), ("clr_max_lr", float), ("clr_stepsize", int), ("clr_gamma", float),
                      ("clr_mode", str), ("one_cycle_annealing_factor", float), ("start_lr", float), ("end_lr", float)):
        ap.add_argument("--" + name, type=typ, default=None)
    FLAGS = ap.parse_args()
    FLAGS.model = model
    if not os.path.exists(FLAGS.list):
        raise ValueError("list path must exist")
    return main(FLAGS)


if __name__ == "__main__":
    parse_and_run("FlowNetS")
