"""FlowNetCSS behind the reference's class surface (/root/reference src/flownet_css/flownet_css.py:
FlowNetCSS(Net).model(inputs, training_schedule, trainable) and .loss); the graph itself
is executed by the HIP engine (src/engine.py)."""
from ..net import Net, Mode
from ..losses import multiscale_loss, fusion_loss


class FlowNetCSS(Net):
    model_name = 'FlowNetCSS'

    def __init__(self, mode=Mode.TRAIN, debug=False, dtype="f32"):
        super(FlowNetCSS, self).__init__(mode=mode, debug=debug, dtype=dtype)

    def loss(self, flow, predictions):
        if self.model_name == 'FlowNet2':
            return fusion_loss(flow, predictions)
        return multiscale_loss(flow, predictions, self.weights, self.model_name,
                               gt_scale=20.0 if self.model_name == 'FlowNetSD' else 0.05)
