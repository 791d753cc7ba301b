"""A few full-catalog scoring calls at Beauty's shape (iid scores): the target of rocprofv3 kernel-trace / --pmc runs of the score path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recboard_amd import ops  # noqa: E402
U, N = 22363, 12101
gq = torch.Generator(device="cuda").manual_seed(11)
q = torch.randn(U, 64, device="cuda", generator=gq)
E = torch.randn(N, 64, device="cuda", generator=gq)
sp = torch.arange(0, U + 1, device="cuda") * 8
si = torch.sort(torch.randint(0, N, (U, 8), device="cuda"), 1).values.reshape(-1)
prep = ops.score_prepare(E)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    ops.score_topk(q, E, sp, si, 50, prep=prep)
torch.cuda.synchronize()
