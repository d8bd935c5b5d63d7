"""SSDLoss -- same constructor and `compute_loss` entry as
localisation_part/keras_loss_function/keras_ssd_loss.py:22-211.  `model.compile(loss=ssd_loss.compute_loss)`
recognises the bound method and lowers it to the fused multibox-loss kernels (dj_ssd_loss_fwd/bwd) inside the
training plan: log-loss on the softmax outputs with the 1e-15 clamp, smooth-L1 on the 4 offsets, batch-wide top-k
hard-negative mining (neg_pos_ratio : 1, at least n_neg_min, at most the non-zero negative losses), normalised by
the number of positive boxes.

Called directly, `compute_loss(y_true, y_pred)` follows the Keras loss protocol on device-resident tensors
(torch CUDA, shape (batch, #boxes, n_classes + 12)) and returns the reference's (batch,) vector
(keras_ssd_loss.py:205-209), computed by the same kernels.  There is no host path: CPU arrays raise."""
import torch


class SSDLoss:
    _dj_loss = "ssd"

    def __init__(self, neg_pos_ratio=3, n_neg_min=0, alpha=1.0):
        self.neg_pos_ratio = neg_pos_ratio
        self.n_neg_min = n_neg_min
        self.alpha = alpha

    def compute_loss(self, y_true, y_pred):
        from ..engine import call, query
        for t in (y_true, y_pred):
            if not (isinstance(t, torch.Tensor) and t.is_cuda):
                raise TypeError("SSDLoss.compute_loss runs on the MI355X: y_true / y_pred must be torch CUDA tensors "
                                "(inside Model.compile it is lowered to the same kernels); there is no CPU path")
        if y_true.shape != y_pred.shape or y_pred.dim() != 3 or y_pred.shape[-1] <= 13:
            raise ValueError("SSDLoss.compute_loss: expected two (batch, #boxes, n_classes + 12) tensors, got %s and %s"
                             % (tuple(y_true.shape), tuple(y_pred.shape)))
        yt = y_true.detach().to(torch.float32).contiguous()
        yp = y_pred.detach().to(torch.float32).contiguous()
        batch, per_image, width = yp.shape
        nbox, n_cls = batch * per_image, width - 12
        ws = torch.empty(query("dj_ssd_loss_workspace_floats", nbox), dtype=torch.float32, device=yp.device)
        out = torch.zeros(8, dtype=torch.float32, device=yp.device)
        call("dj_ssd_loss_fwd", yt, yp, nbox, n_cls, int(self.neg_pos_ratio), int(self.n_neg_min), float(self.alpha),
             ws, out)
        # workspace: cls | loc | positives | negative losses | kept negatives (csrc/dj_loss.hip); the per-image sums of
        # keras_ssd_loss.py:196-209 are views over it
        cls, loc, pos, keep = (ws[i * nbox:(i + 1) * nbox].view(batch, per_image) for i in (0, 1, 2, 4))
        total = (cls * (pos + keep)).sum(dim=1) + float(self.alpha) * (loc * pos).sum(dim=1)
        return total / torch.clamp_min(out[1], 1.0) * float(batch)
