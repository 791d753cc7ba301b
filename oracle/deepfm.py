"""Oracle: DeepFM encode / fit / recommend_from_pool (TEST INFRASTRUCTURE).  torch-CPU fp32.

  LogisticRegression       DeepFM/main.py:34-62    lr = sum_f w_f[x_f] + bias
  InnerProductInteraction  DeepFM/main.py:65-85    fm = 0.5 * sum_d ((sum_f E)^2 - sum_f E^2)
  MLPBlock / dnn           DeepFM/main.py:103-124,151-164   Linear -> BatchNorm1d -> ReLU -> Dropout, then Linear(., 1)
  encode                   DeepFM/main.py:201-209  logits = lr + fm + dnn(flatten E)
  fit                      DeepFM/main.py:211-215  BCELoss4Logits(mean)(logits, labels)
  recommend_from_pool      DeepFM/main.py:217-219  sigmoid(logits)

All fields are EMBED fields here (the benchmark schema Frappe_x1_BARS, DeepFM/configs/Frappe_x1_BARS.yaml).
"""
import torch
import torch.nn.functional as F

from . import criterions


def encode(tables, tables_lr, lr_bias, mlp, x, training, eps=1e-5):
    """tables[f] [count_f, D]; tables_lr[f] [count_f, 1]; x int64 [B, F];
    mlp = list of dicts(linear.weight, linear.bias, [bn.weight, bn.bias, bn.running_mean, bn.running_var]) + final (weight, bias)."""
    B, nf = x.shape
    E = torch.stack([tables[f][x[:, f]] for f in range(nf)], 1)              # [B, F, D]
    lr = torch.stack([tables_lr[f][x[:, f]] for f in range(nf)], 1).sum(1) + lr_bias  # [B, 1]
    fm = 0.5 * (E.sum(1) ** 2 - (E ** 2).sum(1)).sum(-1, keepdim=True)
    h = E.flatten(1)
    for blk in mlp[:-1]:
        h = h @ blk["linear.weight"].T + blk["linear.bias"]
        if "bn.weight" in blk:
            if training:
                mu = h.mean(0)
                var = h.var(0, unbiased=False)
            else:
                mu, var = blk["bn.running_mean"], blk["bn.running_var"]
            h = (h - mu) / torch.sqrt(var + eps) * blk["bn.weight"] + blk["bn.bias"]
        h = torch.relu(h)
    h = h @ mlp[-1]["weight"].T + mlp[-1]["bias"]
    return lr + fm + h


def fit(tables, tables_lr, lr_bias, mlp, x, labels):
    return criterions.bce_with_logits(encode(tables, tables_lr, lr_bias, mlp, x, True), labels)
