"""Cycle accounting of one wave of the score kernel (needs recboard_amd/librecengine_prof.so = a -DSC_PROFILE build of
score.hip: `make -C recboard_amd/csrc prof`): per stage, where the wave's time goes -- exact form vs split form."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "librecengine_prof.so")
from recboard_amd import ops
L = lib.load()
for n, a in (("re_dbg_score_diag", [ctypes.c_int]), ("re_dbg_score_x2", [ctypes.c_int]), ("re_dbg_score_counters", [ctypes.c_void_p, ctypes.c_int]),
             ("re_dbg_score_counters_x", [ctypes.c_void_p])):
    getattr(L, n).argtypes = a; getattr(L, n).restype = None
U, N, D = 22363, 12101, 64
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(U, D, device="cuda", generator=g); E = torch.randn(N, D, device="cuda", generator=g)
sp = torch.arange(0, U + 1, device="cuda") * 8
si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
if os.environ.get("X2_STATE"):   # the bench's trained state, dumped by scripts/x2_bench_state.py (X2_DUMP=...)
    st = torch.load(os.environ["X2_STATE"])
    q, E, sp, si = (st[k].cuda().contiguous() for k in ("q", "E", "sp", "si"))
buf = (ctypes.c_ulonglong * 4)(); bx = (ctypes.c_ulonglong * 2)()
for x2 in (0, 1):
    L.re_dbg_score_x2(x2)
    for mode, name in ((8, "normal"), (9, "no hits")):
        L.re_dbg_score_diag(mode)
        for _ in range(2):
            ops.score_topk(q, E, sp, si, 50); torch.cuda.synchronize()
        L.re_dbg_score_counters(buf, 0); L.re_dbg_score_counters_x(bx)
        n = max(bx[1], 1)
        tot = (buf[0] + buf[1] + buf[2] + buf[3] + bx[0]) / n
        print(f"x2={x2} {name:8s}: per stage (2 tiles) of one wave, {n} stages: barrier-1 wait {buf[0]/n:.0f}  LDS store(+wg drain) {buf[1]/n:.0f}  "
              f"barrier-2 wait {buf[2]/n:.0f}  prefetch+frag+MFMA {buf[3]/n:.0f}  filter/append(+local drain) {bx[0]/n:.0f}  sum {tot:.0f} ticks", flush=True)
L.re_dbg_score_diag(0)
