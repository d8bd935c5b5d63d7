"""keras.regularizers.l2 as used on the SSD head kernels
(localisation_part/models/keras_ssd300_dct_j2d_resnet.py:490...673): penalty l2 * sum(w^2)."""


class L1L2(object):
    def __init__(self, l1=0.0, l2=0.0):
        if l1:
            raise NotImplementedError("l1 regularisation is not on the reference's hot path")
        self.l1 = float(l1)
        self.l2 = float(l2)

    def get_config(self):
        return {"l1": self.l1, "l2": self.l2}


def l2(l=0.01):
    return L1L2(l2=l)
