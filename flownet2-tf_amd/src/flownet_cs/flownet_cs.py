"""FlowNetCS behind the reference's class surface (/root/reference src/flownet_cs/flownet_cs.py:
FlowNetCS(Net).model(inputs, training_schedule, trainable) and .loss); the graph itself
is executed by the HIP engine (src/engine.py)."""
from ..net import Net, Mode
from ..losses import multiscale_loss, fusion_loss


class FlowNetCS(Net):
    model_name = 'FlowNetCS'

    def __init__(self, mode=Mode.TRAIN, debug=False, dtype="f32"):
        super(FlowNetCS, self).__init__(mode=mode, debug=debug, dtype=dtype)

    def loss(self, flow, predictions):
        if self.model_name == 'FlowNet2':
            return fusion_loss(flow, predictions)
        return multiscale_loss(flow, predictions, self.weights, self.model_name,
                               gt_scale=20.0 if self.model_name == 'FlowNetSD' else 0.05)
