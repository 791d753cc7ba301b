import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops
U, N, D = 22363, 12101, 64
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(U, D, device="cuda", generator=g); E = torch.randn(N, D, device="cuda", generator=g)
sp = torch.arange(0, U + 1, device="cuda") * 8
si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
for _ in range(10):
    ops.score_topk(q, E, sp, si, 50)
torch.cuda.synchronize()
for _ in range(10):
    ops.score_topk(q[:512], E, sp[:513], si, 50)
torch.cuda.synchronize()
