"""DispSmoothL1Loss: drop-in for dmb/modeling/stereo/losses/smooth_l1_loss.py:6-95 (masked mean of smooth_l1)."""
from ._common import MapLoss, per_level, scaled_gt


class DispSmoothL1Loss(object):
    def __init__(self, max_disp, start_disp=0, weights=None, sparse=False):
        self.max_disp, self.start_disp, self.weights, self.sparse = max_disp, start_disp, weights, sparse

    def loss_per_level(self, estDisp, gtDisp):
        gt, scale = scaled_gt(gtDisp, estDisp.shape[-2:], self.sparse)
        return MapLoss.apply(estDisp, gt.detach().contiguous(), self.start_disp, self.max_disp / scale, 1)

    def __call__(self, estDisp, gtDisp):
        if not isinstance(estDisp, (list, tuple)):
            estDisp = [estDisp]
        weights = per_level(self.weights, len(estDisp))
        return {"l1_loss_lvl{}".format(i): weights[i] * self.loss_per_level(d, gtDisp) for i, d in enumerate(estDisp)}

    @property
    def name(self):
        return 'SmoothL1Loss'
