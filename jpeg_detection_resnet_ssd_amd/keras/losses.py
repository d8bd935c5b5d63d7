"""Loss functions understood by Model.compile (Keras loss protocol: loss(y_true, y_pred) -> (batch,) tensor)."""
import torch


def categorical_crossentropy(y_true, y_pred):
    """keras.losses.categorical_crossentropy on softmax outputs (classification_part/config/resnet/config_file.py:64).
    Model.compile lowers it to dj_categorical_crossentropy inside the plan; called directly it runs the same kernel on
    device-resident (torch CUDA) tensors and returns the per-sample losses."""
    from ..engine import call
    for t in (y_true, y_pred):
        if not (isinstance(t, torch.Tensor) and t.is_cuda):
            raise TypeError("categorical_crossentropy runs on the MI355X: pass torch CUDA tensors (no CPU path)")
    yt = y_true.detach().to(torch.float32).contiguous()
    yp = y_pred.detach().to(torch.float32).contiguous()
    rows, c = yp.shape[0], yp.shape[-1]
    loss_rows = torch.empty(rows, dtype=torch.float32, device=yp.device)
    out = torch.zeros(8, dtype=torch.float32, device=yp.device)
    call("dj_categorical_crossentropy", yt, yp, rows, c, 1.0, loss_rows, None, out)
    return loss_rows


categorical_crossentropy._dj_loss = "categorical_crossentropy"
