"""rocprofv3 target: the score kernel's fixed per-workgroup cost (start-up + final output) -- tiny catalogs, same users."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops
U, D = 22363, 64
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(U, D, device="cuda", generator=g)
for N in (64, 256, 1024):
    E = torch.randn(N, D, device="cuda", generator=g)
    sp = torch.arange(0, U + 1, device="cuda") * 8
    si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
    for _ in range(5):
        ops.score_topk(q, E, sp, si, 50)
torch.cuda.synchronize()
