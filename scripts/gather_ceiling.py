"""Where is the ceiling for the D=64 gather on this box?  device-to-device copy, gather with sequential / sorted / random indices."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops
def ev(fn, it=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True); a.record()
    for _ in range(it): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) / it
R, n, D = 16 * 1024 * 1024, 4 * 1024 * 1024, 64
W = torch.randn(R, D, device="cuda")
src = torch.randn(n, D, device="cuda"); dst = torch.empty_like(src)
ms = ev(lambda: dst.copy_(src)); print(f"copy 1 GiB -> 1 GiB          : {ms:.4f} ms  {2*n*D*4/ms/1e6:.0f} GB/s (read + write)")
g = torch.Generator(device="cuda").manual_seed(0)
for name, idx in (("sequential rows", torch.arange(n, device="cuda")), ("sorted random rows", torch.sort(torch.randint(0, R, (n,), device="cuda", generator=g)).values),
                  ("random rows", torch.randint(0, R, (n,), device="cuda", generator=g))):
    out = torch.empty(n, D, device="cuda")
    ms = ev(lambda: ops.gather_rows(W, idx))
    print(f"gather {name:22s}: {ms:.4f} ms  {n*520/ms/1e6:.0f} GB/s (8 + 256 + 256 B per row)")
