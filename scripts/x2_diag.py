"""Where the split score kernel's time goes: diagnostic modes of score_kernel_reg (1 = no hits, 2 = append but never insert,
3 = counters) for the exact form and the split form, Beauty shape, iid scores."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops, lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "librecengine_dbg.so")   # the re_dbg_* switches live in the diagnostic twin (make -C recboard_amd/csrc dbg)
L = lib.load()
for n, a in (("re_dbg_score_diag", [ctypes.c_int]), ("re_dbg_score_x2", [ctypes.c_int]), ("re_dbg_score_counters", [ctypes.c_void_p, ctypes.c_int]),
             ("re_dbg_score_vote", [ctypes.c_int])):
    getattr(L, n).argtypes = a; getattr(L, n).restype = None
U, N, D = (int(sys.argv[1]), int(sys.argv[2]), 64) if len(sys.argv) > 2 else (22363, 12101, 64)
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(U, D, device="cuda", generator=g); E = torch.randn(N, D, device="cuda", generator=g)
sp = torch.arange(0, U + 1, device="cuda") * 8
si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
def t(fn, it=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / it
buf = (ctypes.c_ulonglong * 4)()
for x2 in (0, 1):
    L.re_dbg_score_x2(x2)
    for mode, name in ((0, "normal"), (1, "no hits"), (2, "append, never insert")):
        L.re_dbg_score_diag(mode)
        print(f"x2={x2} {name:24s}: {t(lambda: ops.score_topk(q, E, sp, si, 50)):.3f} ms", flush=True)
    L.re_dbg_score_diag(3)
    L.re_dbg_score_counters(buf, 1)
    ops.score_topk(q, E, sp, si, 50); torch.cuda.synchronize()
    L.re_dbg_score_counters(buf, 1)
    nw = 512 * 4
    print(f"x2={x2} one launch: drains/wave {buf[0]/nw:.1f}  rounds/wave {buf[1]/nw:.1f}  hits/lane {buf[2]/nw/64:.1f}")
    L.re_dbg_score_diag(0)
# ablation of score_topk_merge_x (timing only: results are wrong in these modes)
L.re_dbg_score_mxdiag.argtypes = [ctypes.c_int]; L.re_dbg_score_mxdiag.restype = None
L.re_dbg_score_x2(1)
for m, name in ((0, "full"), (1, "no list merge"), (2, "no re-scoring"), (4, "no final sort"), (7, "none of the three")):
    L.re_dbg_score_mxdiag(m)
    print(f"merge_x {name:20s}: whole call {t(lambda: ops.score_topk(q, E, sp, si, 50)):.3f} ms", flush=True)
L.re_dbg_score_mxdiag(0)
