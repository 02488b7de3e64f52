"""python -m src.flownet_css.train --list train.txt --out ./logs [...]: the reference's src/flownet_css/train.py + Net.train
(net.py:1002-1400) for FlowNetCSS over the HIP trainer -- the flags and the data path of src.flownet_s.train; what is trained is
what the reference's graph leaves trainable: the last FlowNetS (the networks in front are built trainable=False, flownet_css.py:18)."""
from ..flownet_s.train import parse_and_run

if __name__ == "__main__":
    parse_and_run("FlowNetCSS")
