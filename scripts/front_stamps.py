"""Phase stamps of score_front_k's first 256 workgroups (debug library): python scripts/front_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recboard_amd import lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "librecengine_dbg.so")
from recboard_amd import ops
U, N, D = 22363, 12101, 64
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(U, D, device="cuda", generator=g); E = torch.randn(N, D, device="cuda", generator=g)
sp = torch.arange(0, U + 1, device="cuda") * 8
si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
for _ in range(5):
    ops.score_topk(q, E, sp, si, 50)
torch.cuda.synchronize()
L = lib.load()
buf = (ctypes.c_ulonglong * (256 * 16))()
L.re_dbg_front_stamps.argtypes = [ctypes.c_void_p]; L.re_dbg_front_stamps.restype = ctypes.c_int
assert L.re_dbg_front_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16).astype(np.int64)
t0 = a[:, 0].min()
rel = (a - t0)
print("per phase (cycles of the 100 MHz / shader clock counter), median over 256 workgroups; first row = start offsets")
names = ["start", "Q split done", "tile 0", "tile 1", "tile 2", "tile 3", "tile 4", "tile 5", "-", "-", "tiles done", "end"]
for i, nme in enumerate(names):
    col = rel[:, i]
    print(f"{nme:14s} min {col.min():9d}  median {int(np.median(col)):9d}  max {col.max():9d}")
w = (ctypes.c_ulonglong * (1024 * 2))()
L.re_dbg_front_wall.argtypes = [ctypes.c_void_p]; L.re_dbg_front_wall.restype = ctypes.c_int
assert L.re_dbg_front_wall(w) == 0
ww = np.frombuffer(w, dtype=np.uint64).reshape(1024, 2).astype(np.int64)
nb = int((ww[:, 0] > 0).sum())
ww = ww[:nb]
t0 = ww[:, 0].min()
st, en = (ww[:, 0] - t0) / 100.0, (ww[:, 1] - t0) / 100.0     # us
print(f"{nb} workgroups; start us: min {st.min():.2f} median {np.median(st):.2f} max {st.max():.2f}; end us: min {en.min():.2f} median {np.median(en):.2f} max {en.max():.2f}")
for lo, hi, nme in ((0, 63, "split blocks"), (63, nb, "bound blocks")):
    print(f"  {nme}: start {st[lo:hi].min():.2f}..{st[lo:hi].max():.2f}  end {en[lo:hi].min():.2f}..{en[lo:hi].max():.2f}  lifetime median {np.median(en[lo:hi]-st[lo:hi]):.2f} max {(en[lo:hi]-st[lo:hi]).max():.2f}")
late = np.sort(st[63:])
print("bound blocks starting after 1 us:", int((late > 1.0).sum()), "of", late.size, "; start-time deciles:", np.round(np.percentile(late, [10, 30, 50, 70, 80, 90, 95, 100]), 2))
