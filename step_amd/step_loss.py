"""Loss hook with the reference's signature (``step/step_loss/step_loss.py:5-16`` +
``basicts/metrics/mae.py:5-28``): ``step_loss(prediction, real_value, theta, priori_adj,
gsl_coefficient, null_val)``.  The reference runner calls it on RESCALED predictions
(``base_tsf_runner.py:240-250``); it is caller-side code, a handful of element-wise torch ops on
``[B,12,N,1]`` / ``[B,N,N]`` tensors whose autograd feeds ``dpred`` / ``dtheta`` into the native backward.
"""
import numpy as np
import torch


def masked_mae(preds, labels, null_val=np.nan):
    if np.isnan(null_val):
        mask = ~torch.isnan(labels)
    else:
        mask = (labels - null_val).abs() > 5e-5          # ~isclose(labels, null_val, atol=5e-5, rtol=0)
    mask = mask.float()
    mask = mask / torch.mean(mask)
    mask = torch.nan_to_num(mask, nan=0.0)
    loss = torch.abs(preds - labels) * mask
    return torch.mean(torch.nan_to_num(loss, nan=0.0))


def step_loss(prediction, real_value, theta, priori_adj, gsl_coefficient, null_val=np.nan):
    B, N, _ = theta.shape
    t = theta.reshape(B, N * N)
    y = priori_adj.reshape(B, N * N)
    loss_graph = torch.nn.functional.binary_cross_entropy(t, y)
    loss_pred = masked_mae(prediction, real_value, null_val)
    return loss_pred + loss_graph * gsl_coefficient
