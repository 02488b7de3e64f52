"""FlowNetS_interp behind the reference's class surface (/root/reference
src/flownet_s_interp/flownet_s_interp.py:10-254): the FlowNetS tower fed with the first image, the sparse flow
of a set of matches (scaled by 0.05) and the match mask -- flow interpolation instead of flow estimation.
Variable scope stays 'FlowNetS' (:23), so FlowNetS checkpoints load; with no_deconv_biases (the class default)
the predict_flow layers carry no biases (:86-95).  The graph runs on the HIP engine (src/engine.py)."""
from ..net import Net, Mode
from .. import weights as W
from ..losses import multiscale_hfem_loss


class FlowNetS_interp(Net):
    model_name = 'FlowNetS_interp'
    scope = 'FlowNetS'

    def __init__(self, mode=Mode.TRAIN, debug=False, no_deconv_biases=True, dtype="f32"):
        super(FlowNetS_interp, self).__init__(mode=mode, debug=debug, dtype=dtype)
        self.no_deconv_biases = no_deconv_biases

    def _init_weights(self, seed):
        return W.init_weights(self.model_name, seed, head_biases=not self.no_deconv_biases)

    # A plain FlowNetS checkpoint carries head biases: with no_deconv_biases they are not variables of this graph
    # and the engine ignores them (Engine.ignored_variables), as the reference's Saver would.

    def _engine_kwargs(self):
        return {"no_deconv_biases": self.no_deconv_biases}

    def model(self, inputs, training_schedule=None, trainable=True, is_training=True):
        """inputs: {'input_a' [N,H,W,3], 'matches_a' [N,H,W,1], 'sparse_flow' [N,H,W,2]}; is_training=False returns
        only {'flow'} like the reference (:146-156)."""
        a = inputs['input_a']
        n, h, w, _ = a.shape
        eng = self.engine(int(n), int(h), int(w))
        eng.set_inputs_interp(a, inputs['matches_a'], inputs['sparse_flow'])
        eng.launch()
        out = {k: v.clone() for k, v in eng.outputs.items()}
        return out if is_training else {'flow': out['flow']}

    def loss(self, targets, predictions, add_hard_flow_mining='', lambda_weight=2., hard_examples_perc=50, edges=None):
        return multiscale_hfem_loss(targets, predictions, add_hard_flow_mining, lambda_weight, hard_examples_perc,
                                    edges, self.weights, self.scope)
