"""SSDLoss -- same constructor and `compute_loss` entry as
localisation_part/keras_loss_function/keras_ssd_loss.py:22-211.  `model.compile(loss=ssd_loss.compute_loss)`
recognises the bound method and lowers it to the fused multibox-loss kernels (dj_ssd_loss_fwd/bwd):
log-loss on the softmax outputs with the 1e-15 clamp, smooth-L1 on the 4 offsets, batch-wide top-k
hard-negative mining (neg_pos_ratio : 1, at least n_neg_min, at most the non-zero negative losses),
normalised by the number of positive boxes."""


class SSDLoss:
    _dj_loss = "ssd"

    def __init__(self, neg_pos_ratio=3, n_neg_min=0, alpha=1.0):
        self.neg_pos_ratio = neg_pos_ratio
        self.n_neg_min = n_neg_min
        self.alpha = alpha

    def compute_loss(self, y_true, y_pred):
        raise RuntimeError("SSDLoss.compute_loss is lowered by Model.compile to the HIP multibox loss; it is not "
                           "callable on host arrays (there is no CPU path)")
