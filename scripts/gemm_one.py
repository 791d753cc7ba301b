"""Ten launches of one product (scripts/pmc_gemm.sh): python scripts/gemm_one.py [M N K tA tB]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops
a = sys.argv[1:]
M, N, K = (int(v) for v in a[:3]) if len(a) >= 3 else (4096, 400, 4096)
tA, tB = (bool(int(a[3])), bool(int(a[4]))) if len(a) >= 5 else (False, True)
A = torch.randn((K, M) if tA else (M, K), device="cuda")
B = torch.randn((N, K) if tB else (K, N), device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(10):
    ops.gemm(A, B, transA=tA, transB=tB, out=out)
torch.cuda.synchronize()
