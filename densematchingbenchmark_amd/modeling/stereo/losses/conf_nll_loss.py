"""ConfidenceNllLoss: drop-in for dmb/modeling/stereo/losses/conf_nll_loss.py:6-85 (masked mean of -logsigmoid)."""
from ._common import MapLoss, per_level, scaled_gt


class ConfidenceNllLoss(object):
    def __init__(self, max_disp, start_disp=0, weights=None, sparse=False):
        self.max_disp, self.start_disp, self.weights, self.sparse = max_disp, start_disp, weights, sparse

    def loss_per_level(self, estConf, gtDisp):
        gt, scale = scaled_gt(gtDisp, estConf.shape[-2:], self.sparse)
        return MapLoss.apply(estConf, gt.detach().contiguous(), self.start_disp, self.max_disp / scale, 0)

    def __call__(self, estConf, gtDisp):
        if not isinstance(estConf, (list, tuple)):
            estConf = [estConf]
        weights = per_level(self.weights, len(estConf))
        return {"conf_loss_lvl{}".format(i): weights[i] * self.loss_per_level(c, gtDisp) for i, c in enumerate(estConf)}

    @property
    def name(self):
        return 'ConfidenceNLLLoss'
