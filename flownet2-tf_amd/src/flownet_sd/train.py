"""python -m src.flownet_sd.train --list train.txt|x.tfrecords --out ./logs [...]: the reference's src/flownet_sd/train.py
(+ Net.train) over the HIP trainer -- FlowNetSD has neither correlation nor flow_warp, so it shares FlowNetS's training
path (src/flownet_s/train.py: data pipeline, multiscale EPE loss with labels 20 * gt as flownet_sd.py:122, Adam on
LONG_SCHEDULE, data-parallel all-reduce, .npz or TensorFlow checkpoints under the reference's variable names)."""
from ..flownet_s.train import parse_and_run

if __name__ == "__main__":
    parse_and_run("FlowNetSD")
