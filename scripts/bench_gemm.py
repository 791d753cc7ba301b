"""GEMM kernel timings on the shapes the engines use (DeepFM MLP fwd/bwd, SASRec CE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops

def ev(fn, iters=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters * 1e3

g = torch.Generator(device="cuda").manual_seed(0)
shapes = [("DeepFM fwd  h W^T", 4096, 400, 400, False, True), ("DeepFM fwd  x W0^T", 4096, 400, 100, False, True),
          ("DeepFM bwd  dz W", 4096, 400, 400, False, False), ("DeepFM bwd  dz^T h", 400, 400, 4096, True, False),
          ("CE logits u E^T", 3031, 12101, 64, False, True), ("CE dU = dlogits E", 3031, 64, 12101, False, False),
          ("CE dE = dlogits^T u", 12101, 64, 3031, True, False), ("square", 8192, 8192, 512, False, True)]
for name, M, N, K, tA, tB in shapes:
    A = torch.randn((K, M) if tA else (M, K), device="cuda", generator=g)
    Bm = torch.randn((N, K) if tB else (K, N), device="cuda", generator=g)
    out = torch.empty(M, N, device="cuda")
    us = ev(lambda: ops.gemm(A, Bm, transA=tA, transB=tB, out=out))
    ref = (A.T if tA else A).double() @ (Bm.T if tB else Bm).double()
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    print(f"{name:22s} M={M:6d} N={N:6d} K={K:6d}: {us:8.1f} us  {2*M*N*K/us/1e6:6.1f} TF   rel err {err:.1e}")
