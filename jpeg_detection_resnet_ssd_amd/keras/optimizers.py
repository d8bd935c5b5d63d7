"""keras.optimizers.SGD with the Keras 2.2.4 update rule (differs from torch.optim.SGD whenever the
learning rate changes): lr_t = lr / (1 + decay * iterations); v = momentum*v - lr_t*g;
p += v  (or p += momentum*v - lr_t*g with nesterov=True).
Call sites: localisation_part/training_dct_pascal_j2d_resnet.py:152,
classification_part/config/resnet/config_file.py:58-63."""


class Optimizer(object):
    pass


class SGD(Optimizer):
    def __init__(self, lr=0.01, momentum=0.0, decay=0.0, nesterov=False, **kwargs):
        self.lr = float(lr)
        self.momentum = float(momentum)
        self.decay = float(decay)
        self.initial_decay = float(decay)
        self.nesterov = bool(nesterov)
        self.iterations = 0

    def current_lr(self):
        lr = self.lr
        if self.initial_decay > 0:
            lr = lr * (1.0 / (1.0 + self.decay * self.iterations))
        return lr

    def get_config(self):
        return {"lr": self.lr, "momentum": self.momentum, "decay": self.decay, "nesterov": self.nesterov}
