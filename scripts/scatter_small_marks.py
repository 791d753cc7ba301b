"""Shader-clock stamps of workgroup 0 of re_scatter_add_rows_small.  Needs the diagnostic build:
    cd recboard_amd/csrc && hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -I../../include -DSO_MARKS -c scatter.hip -o /tmp/scatter_m.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../var_somarks.so $(ls build/*.o | grep -v scatter.o) /tmp/scatter_m.o"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recboard_amd import lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "var_somarks.so")
import numpy as np, torch
from recboard_amd import ops
R, D, NR = 12102, 64, 32768
g = torch.Generator(device="cuda").manual_seed(1)
G = torch.randn(3, NR, D, device="cuda", generator=g)
out = torch.empty(R, D, device="cuda")
w = 1.0 / np.arange(1, R); w /= w.sum()
rng = np.random.default_rng(0)
for name, n, mk in (("empty", 4496, lambda n: np.zeros((3, n), np.int64)), ("uniform", 4496, lambda n: rng.integers(1, R, (3, n))),
                    ("zipf (hot row 1 is NOT workgroup 0's)", 4496, lambda n: rng.choice(R - 1, (3, n), p=w) + 1),
                    ("zipf shifted: hot row 256 -> workgroup 0", 4496, lambda n: (rng.choice(R - 1, (3, n), p=w) + 255) % (R - 1) + 1)):
    keys = torch.zeros(3, NR, dtype=torch.int32, device="cuda")
    keys[:, :n] = torch.from_numpy(mk(n).astype(np.int32)).cuda()
    for _ in range(3):
        ops.scatter_add_rows_small(G, keys, R, out, n_regions=3, region_stride=NR, n=n)
    torch.cuda.synchronize()
    t = out[0, :11].cpu().numpy().astype(np.int64)
    print(name, "stamps (ticks from start): init", t[1], "chunks", t[2:8], "scan end", t[8], "flush", t[9], "final", t[10])
