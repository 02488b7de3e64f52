"""FlowNet2 behind the reference's class surface (/root/reference src/flownet2/flownet2.py:
FlowNet2(Net).model(inputs, training_schedule, trainable) and .loss); the graph itself
is executed by the HIP engine (src/engine.py)."""
from ..net import Net, Mode
from ..losses import multiscale_loss, fusion_loss


class FlowNet2(Net):
    model_name = 'FlowNet2'

    def __init__(self, mode=Mode.TRAIN, debug=False, dtype="f32"):
        super(FlowNet2, self).__init__(mode=mode, debug=debug, dtype=dtype)

    def loss(self, flow, predictions):
        if self.model_name == 'FlowNet2':
            return fusion_loss(flow, predictions, self.weights, 'FlowNet2')
        return multiscale_loss(flow, predictions, self.weights, self.model_name,
                               gt_scale=20.0 if self.model_name == 'FlowNetSD' else 0.05)
