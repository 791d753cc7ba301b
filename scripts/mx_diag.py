"""Where score_topk_merge_x's time goes: the whole re_score_topk call (Beauty shape, iid scores) with parts of the merge switched off in the
diagnostic library (re_dbg_score_mxdiag: 1 = only the first round of list merging, 2 = no exact re-scoring, 4 = no final sort; results are wrong
with any of them -- timing only)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "librecengine_dbg.so")
L = lib.load()
from recboard_amd import ops
U, N = 22363, 12101
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(U, 64, device="cuda", generator=g); E = torch.randn(N, 64, device="cuda", generator=g)
sp = torch.arange(0, U + 1, device="cuda") * 8
si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
L.re_dbg_score_mxdiag.argtypes = [ctypes.c_int]
for m in (0, 1, 2, 4, 3, 7, 0):
    L.re_dbg_score_mxdiag(m)
    for _ in range(3):
        ops.score_topk(q, E, sp, si, 50)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.score_topk(q, E, sp, si, 50)
    e1.record(); torch.cuda.synchronize()
    print(f"mxdiag {m}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call", flush=True)
