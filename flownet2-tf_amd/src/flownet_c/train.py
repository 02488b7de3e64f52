"""python -m src.flownet_c.train --list train.txt --out ./logs --dtype f32 [...]: the reference's src/flownet_c/train.py +
Net.train (net.py:1002-1400) for FlowNetC over the HIP trainer -- the flags and the data path of src.flownet_s.train.  The
whole network trains: one set of conv1-3 variables for both towers (reuse=True, flownet_c.py:34-37), gradients through the
correlation (CorrelationGrad, src/correlation.py:17-35).  fp32 only (--dtype f32): the correlation gradient op is fp32."""
from ..flownet_s.train import parse_and_run

if __name__ == "__main__":
    parse_and_run("FlowNetC")
