"""Full-catalog score + top-50 at D = 128 (config 5's evaluation shape per GPU: one 512-user batch against a 12.5 M-item shard)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops
def t(fn, it=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / it
for U, N, D in ((512, 12_500_000, 128), (512, 12_500_000, 64), (4096, 1_000_000, 128)):
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn(U, D, device="cuda", generator=g); E = torch.randn(N, D, device="cuda", generator=g)
    sp = torch.arange(0, U + 1, device="cuda") * 8
    si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
    ms = t(lambda: ops.score_topk(q, E, sp, si, 50))
    fl = 2 * D * U * N
    print(f"{U} x {N} D={D}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s ({fl/ms/1e9/157.3*100:.0f} % of fp32 MFMA peak)  table read {N*D*4/ms/1e6:.0f} GB/s")
    prep = ops.score_prepare(E)   # the table's bf16 planes built once (an evaluation scores many user batches against one table)
    ms = t(lambda: ops.score_topk(q, E, sp, si, 50, prep=prep))
    print(f"   with the table prepared once: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s ({fl/ms/1e9/157.3*100:.0f} % of fp32 MFMA peak)")
    del prep
    del E
