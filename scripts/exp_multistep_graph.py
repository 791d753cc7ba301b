"""Timing experiment: k whole steps (batch preparation launch included, on fixed batches) captured as ONE graph -- what does the end of a graph
cost per step?  The step scalars are the captured ones (timing only, not a training run)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from recboard_amd import ops
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in bench.synth_batches(cfg, 8, 1)]
B, S = 512, 50
for k in (0, 1, 2, 4, 8):
    m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
    for i in range(10):
        m.train_step_graph(*bs[i % 8])
    torch.cuda.synchronize()
    if k == 0:
        def run(n):
            for i in range(n):
                m.train_step_graph(*bs[i % 8])
        per = 1
    else:
        g = m._graphs[(B, S, "adam", True)]

        def steps():
            for j in range(k):
                ops.sasrec_batch_prep(*bs[j % 8], blob=g["blob"], state=g["state"], seed=5 + j, step=20 + j, lr=m.lr, beta1=m.betas[0], beta2=m.betas[1],
                                      max_tiles=m._max_tiles(), split=m._split(), tile=m._wave_step(), ncu=m._plan_ncu(), weights=m._prep_weights(B, S))
                m._step_body(g["pb"], 0, seed_dev=g["state"], adam_hyper=g["hyper"])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            steps()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        G = torch.cuda.CUDAGraph()
        with torch.cuda.graph(G, capture_error_mode="thread_local"):
            steps()

        def run(n):
            for i in range(n // k):
                G.replay()
        per = k
    run(40)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        run(320)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 320)
    print(f"steps per graph {k} (0 = the shipped step: preparation launch + graph): {best * 1e6:.1f} us/step", flush=True)
