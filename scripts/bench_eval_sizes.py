"""Full-catalog score + seen-mask + top-50 at the BASELINE catalog sizes (iid scores): Beauty 22 363 x 12 101, Yelp 77 277 x 45 638,
Games 94 762 x 25 612, and one 512-user batch against a 12.5 M-item shard (config 5 per GPU at 8 GPUs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops

def ev(fn, iters=5):
    fn(); fn()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(iters): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters

g = torch.Generator(device="cuda").manual_seed(1)
for name, U, N, D in (("Beauty", 22363, 12101, 64), ("Yelp2018", 77277, 45638, 64), ("Games", 94762, 25612, 64), ("C5 shard, one batch", 512, 12_500_000, 64)):
    q = torch.randn(U, D, device="cuda", generator=g); E = torch.randn(N, D, device="cuda", generator=g)
    sp = torch.arange(0, U + 1, device="cuda") * 8
    si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
    ms = ev(lambda: ops.score_topk(q, E, sp, si, 50), iters=3 if U * N > 2e9 else 10)
    print(f"{name:22s} {U:6d} users x {N:9d} items: {ms:9.3f} ms  {U*N/ms/1e6:8.1f} G items/s  {2*D*U*N/ms/1e9:6.1f} TFLOP/s ({2*D*U*N/ms/1e9/157.3*100:.0f} % of fp32 MFMA peak)")
