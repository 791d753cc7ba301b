"""A / B of the score call between two builds of the library, same box, alternating processes:  python scripts/score_ab.py libA.so libB.so
(child mode: python scripts/score_ab.py --child lib.so -> launch-timed ms of re_score_topk at Beauty's shape, trained-like and iid scores)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "--child":
    from recboard_amd import lib
    lib.LIB_PATH = os.path.join(ROOT, "recboard_amd", sys.argv[2])
    import torch, bench
    from recboard_amd import ops
    U, N = 22363, 12101
    g = torch.Generator(device="cuda").manual_seed(11)
    # (the bench's score leg: bench.roofline_score builds its queries / items the same way)
    r = bench.score_roofline(bench.BEAUTY) if hasattr(bench, "score_roofline") else None
    if r is None:
        q = torch.randn(U, 64, device="cuda", generator=g); E = torch.randn(N, 64, device="cuda", generator=g)
        sp = torch.arange(0, U + 1, device="cuda") * 8
        si = torch.sort(torch.randint(0, N, (U, 8), device="cuda"), 1).values.reshape(-1)
        for _ in range(5): ops.score_topk(q, E, sp, si, 50)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(30): ops.score_topk(q, E, sp, si, 50)
        e1.record(); torch.cuda.synchronize()
        print(f"{sys.argv[2]} iid {e0.elapsed_time(e1) / 30:.4f} ms")
    else:
        print(sys.argv[2], r.get("launch_ms"), r.get("frac"))
    sys.exit(0)
for rep in range(3):
    for so in sys.argv[1:3]:
        print(subprocess.run([sys.executable, __file__, "--child", so], capture_output=True, text=True).stdout.strip().splitlines()[-1], flush=True)
