"""keras.metrics.top_k_categorical_accuracy as the reference's configs use it
(classification_part/config/resnet/config_file.py:19-22): evaluated with torch on the device-resident batch."""
import torch


def top_k_categorical_accuracy(y_true, y_pred, k=5):
    y_true = torch.as_tensor(y_true)
    y_pred = torch.as_tensor(y_pred).to(y_true.device)
    target = y_true.argmax(dim=-1)
    topk = y_pred.topk(k, dim=-1).indices
    return float((topk == target.unsqueeze(-1)).any(dim=-1).float().mean())


def categorical_accuracy(y_true, y_pred):
    return top_k_categorical_accuracy(y_true, y_pred, 1)
