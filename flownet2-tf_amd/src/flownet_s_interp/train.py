"""python -m src.flownet_s_interp.train --records fc_train_all.tfrecords --out ./logs [--steps N --batch 8
                                        --add_hard_flow_mining hard|edges --lambda_weight 2 --hard_examples_perc 50]
The reference's src/flownet_s_interp/train.py + Net.train for the interpolation network over the HIP trainer: the
FlowNetS tower on [image | 0.05 * sparse flow | match mask], heads without biases, the multiscale loss with optional
hard-flow-example mining (flownet_s_interp.py:159-254), Adam on LONG_SCHEDULE, checkpoints (weights + Adam slots)
under the reference's variable names.  Input: the reference's `image_matches` TFRecords (image_a, matches_a,
sparse_flow, edges_a, flow); the interpolation-specific augmentation of the reference is not built."""
import argparse
import os
import time

from ..dataloader import load_interp_batches
from ..flownet_s.train import load_full_checkpoint, save_checkpoint, unpack_weights
from ..training_schedules import LONG_SCHEDULE


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
    wts = W.load_weights(flags.checkpoint) if flags.checkpoint else W.init_weights("FlowNetS_interp", flags.seed)
    for k in [k for k in wts if "/predict_flow" in k and k.endswith("/biases")]:
        del wts[k]  # no_deconv_biases (flownet_s_interp.py:78-95): a FlowNetS checkpoint's head biases are no variables here
    tr = FlowNetSTrainer(wts, flags.batch, flags.height, flags.width, schedule=LONG_SCHEDULE, dtype=flags.dtype,
                         model="FlowNetS_interp", add_hard_flow_mining=flags.add_hard_flow_mining,
                         lambda_weight=flags.lambda_weight, hard_examples_perc=flags.hard_examples_perc)
    if flags.checkpoint:
        tr.load_optimizer_state(load_full_checkpoint(flags.checkpoint))
    os.makedirs(flags.out, exist_ok=True)
    step0, t0 = tr.step_count, time.perf_counter()
    end = step0 + flags.steps
    batches = load_interp_batches(flags.records, flags.batch, (flags.height, flags.width), seed=flags.seed + rank)
    for step, (img, matches, sparse, edges, flow) in enumerate(batches, step0 + 1):
        loss = tr.forward_backward_interp(img, matches, sparse, flow, edges=edges, reduce=True)
        tr.apply_gradients(reduced_world=tr.wait_reduction())
        if rank == 0 and (step % flags.log_every == 0 or step == end):
            print("global step %6d | loss %.5f | %.1f samples/s" % (step, float(loss.item()), world * flags.batch *
                                                                    (step - step0) / (time.perf_counter() - t0)), flush=True)
        if rank == 0 and (step % flags.save_every == 0 or step == end):
            save_checkpoint(flags.out, step, dict(unpack_weights(tr), **tr.optimizer_state()), flags.ckpt_format,
                            stem="flownet_s_interp")
        if step >= end:
            break
    return tr


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", required=True, help=".tfrecords with image_a, matches_a, sparse_flow, edges_a, flow")
    ap.add_argument("--out", required=True)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--ckpt-format", default="npz", choices=["npz", "tf"])
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--dtype", default="f16x2", choices=["f32", "f16x2"])
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--add_hard_flow_mining", default="", choices=["", "hard", "edges"])
    ap.add_argument("--lambda_weight", type=float, default=2.0)
    ap.add_argument("--hard_examples_perc", type=float, default=50)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--log-every", type=int, default=10)
    ap.add_argument("--save-every", type=int, default=1000)
    FLAGS = ap.parse_args()
    if not os.path.exists(FLAGS.records):
        raise ValueError("records path must exist")
    main(FLAGS)
