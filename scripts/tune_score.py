import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops, lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "librecengine_dbg.so")   # the re_dbg_* switches live in the diagnostic twin (make -C recboard_amd/csrc dbg)
import ctypes
L = lib.load(); L.re_dbg_score_variant.argtypes = [ctypes.c_int, ctypes.c_int64]; L.re_dbg_score_variant.restype = None
U, N, D = 22363, 12101, 64
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(U, D, device="cuda", generator=g); E = torch.randn(N, D, device="cuda", generator=g)
sp = torch.arange(0, U + 1, device="cuda") * 8
si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
def t(fn, it=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / it

import bench
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
model = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5).eval()
import numpy as np
eval_seq = torch.from_numpy(np.concatenate([b[0] for b in bench.synth_batches(cfg, 44, 99)])[:U]).cuda()
with torch.no_grad():
    q2 = torch.cat([model.encode(eval_seq[i:i + 512])[0][:, -1, :] for i in range(0, U, 512)]).contiguous()
E2 = model.params["Item.embeddings.weight"].detach()[1:]
for rnd in range(2):
    for name, (qq, EE) in {"random-normal": (q, E), "bench (LN-encoded q, xavier E)": (q2, E2)}.items():
        for pop in (2, 3):
            for minseg in (1,):
                L.re_dbg_score_variant(pop, minseg)
                ms = t(lambda: ops.score_topk(qq, EE, sp, si, 50))
                ms5 = t(lambda: ops.score_topk(qq[:512], EE, sp[:513], si, 50))
                print(f"{name:32s} pop={pop} minseg={minseg:2d}: full {ms:.3f} ms ({2*D*U*N/ms/1e9:.1f} TF)   B=512: {ms5:.3f} ms")

L.re_dbg_score_diag.argtypes = [ctypes.c_int]; L.re_dbg_score_diag.restype = None
L.re_dbg_score_variant(3, 1)
for mode, name in ((0, "normal"), (1, "no hits (MFMA loop + filter only)"), (2, "append every score, never insert")):
    L.re_dbg_score_diag(mode)
    ms = t(lambda: ops.score_topk(q, E, sp, si, 50))
    print(f"reg-list diag {name:40s}: {ms:.3f} ms")
L.re_dbg_score_diag(0)

L.re_dbg_score_counters.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.re_dbg_score_counters.restype = None
buf = (ctypes.c_ulonglong * 4)()
L.re_dbg_score_diag(3)
L.re_dbg_score_counters(buf, 1)
ops.score_topk(q, E, sp, si, 50); torch.cuda.synchronize()
L.re_dbg_score_counters(buf, 1)
nw = 512 * 4
print(f"one launch: drains/wave {buf[0]/nw:.1f}  rounds/wave {buf[1]/nw:.1f}  hits/lane {buf[2]/nw/64:.1f}  (segments/wave 1; MFMA tiles/wave ~130)")
L.re_dbg_score_diag(0)

L.re_dbg_score_maxwgs.argtypes = [ctypes.c_int64]; L.re_dbg_score_maxwgs.restype = None
for nw in (256, 512):
    L.re_dbg_score_maxwgs(nw)
    for mode, name in ((0, "normal"), (5, "no hits")):
        L.re_dbg_score_diag(mode)
        ms = t(lambda: ops.score_topk(q, E, sp, si, 50))
        print(f"workgroups {nw}: {name:10s} {ms:.3f} ms")
L.re_dbg_score_diag(0); L.re_dbg_score_maxwgs(0)

if not hasattr(L, "re_dbg_score_counters_x"):
    sys.exit(0)   # (cycle accounting needs a -DSC_PROFILE build)
L.re_dbg_score_counters_x.argtypes = [ctypes.c_void_p]; L.re_dbg_score_counters_x.restype = None
bx = (ctypes.c_ulonglong * 2)()
for nw in (256, 512):
    L.re_dbg_score_maxwgs(nw)
    for mode, name in ((8, "normal"), (9, "no hits")):
        L.re_dbg_score_diag(mode)
        ops.score_topk(q, E, sp, si, 50); torch.cuda.synchronize()
        L.re_dbg_score_counters(buf, 0); L.re_dbg_score_counters_x(bx)
        n = max(bx[1], 1)
        print(f"workgroups {nw} {name:8s}: per stage (2 tiles) of one wave, {n} stages: barrier-1 wait {buf[0]/n:.0f}  LDS store(+wg drain) {buf[1]/n:.0f}  "
              f"barrier-2 wait {buf[2]/n:.0f}  prefetch+MFMA {buf[3]/n:.0f}  filter/append(+local drain) {bx[0]/n:.0f}")
L.re_dbg_score_diag(0); L.re_dbg_score_maxwgs(0)
