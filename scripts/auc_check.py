import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from recboard_amd import ops
from freerec import metrics
g = torch.Generator(device="cuda").manual_seed(0)
for n in (57, 200, 256, 1000, 4096):
    p = torch.rand(n, device="cuda", generator=g)
    y = (torch.rand(n, device="cuda", generator=g) < 0.5).float()
    a = float(ops.auc(p.contiguous(), y))
    b = float(metrics.auroc(p, y))
    l1 = float(ops.bce_logits(torch.logit(p).contiguous(), y)[0]); l2 = float(metrics.log_loss(p, y))
    print(n, a, b, abs(a - b), "logloss", l1, l2)
