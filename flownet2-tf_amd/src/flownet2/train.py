"""python -m src.flownet2.train --list train.txt --out ./logs [...]: the reference's src/flownet2/train.py + Net.train
(net.py:1002-1400) for FlowNet2 over the HIP trainer -- the flags and the data path of src.flownet_s.train; what is trained is
what the reference's graph leaves trainable: the fusion network under FlowNet2.loss (CSS and SD are built trainable=False, flownet2.py:22-23)."""
from ..flownet_s.train import parse_and_run

if __name__ == "__main__":
    parse_and_run("FlowNet2")
