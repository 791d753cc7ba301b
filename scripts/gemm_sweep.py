"""How the wide fp32 product's time splits into fixed and per-k cost: the DeepFM shapes at growing K (scripts/gemm_sweep.py [--reps 50])."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops


def timed(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    dev = "cuda"
    for name, M, N, tA, tB in (("fwd x W^T", 4096, 400, False, True), ("dx dz W", 4096, 400, False, False), ("dW dz^T x", 400, 400, True, False)):
        for K in (128, 400, 800, 1600, 3200, 4096, 8192):
            A = torch.randn((K, M) if tA else (M, K), device=dev)
            B = torch.randn((N, K) if tB else (K, N), device=dev)
            out = torch.empty(M, N, device=dev)
            us = timed(lambda: ops.gemm(A, B, transA=tA, transB=tB, out=out), a.reps)
            print(f"{name:10s} M={M} N={N} K={K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
