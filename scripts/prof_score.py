"""rocprofv3 target: full-catalog score+top-K launches.  SCORE_DIAG=<mode> selects a diagnostic mode of the kernel
(1 = no hits, 6 = no threshold filter, see csrc/score.hip)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops, lib
U, N, D = 22363, 12101, 64
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(U, D, device="cuda", generator=g); E = torch.randn(N, D, device="cuda", generator=g)
sp = torch.arange(0, U + 1, device="cuda") * 8
si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
mode = int(os.environ.get("SCORE_DIAG", "0"))
L = lib.load()
L.re_dbg_score_diag.argtypes = [ctypes.c_int]; L.re_dbg_score_diag.restype = None
L.re_dbg_score_diag(mode)
for _ in range(10):
    ops.score_topk(q, E, sp, si, 50)
torch.cuda.synchronize()
