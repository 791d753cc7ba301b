"""Timing experiment: how long would a step be with the NEXT batch's preparation launch as a third branch inside the step's graph
(and nothing in front of the replay)?  The staged batch is reused -- timing only, the results are not a training run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from recboard_amd import ops
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in bench.synth_batches(cfg, 8, 1)]
B, S = 512, 50


def run(mode):
    m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
    if mode != "base":
        m.prep_weights_in_batch_prep = False          # (the fragments cannot come from a preparation launch that runs a step early)
        blob2 = torch.zeros(ops.prep_layout(B, S)[1], dtype=torch.uint8, device="cuda")
        st2 = torch.zeros(4, dtype=torch.int32, device="cuda")
        third = torch.cuda.Stream()
        orig = m._step_body

        def body(pb, sd, seed_dev=None, adam_hyper=None):
            main = torch.cuda.current_stream()
            if mode == "third":
                third.wait_stream(main)
                with torch.cuda.stream(third):
                    ops.sasrec_batch_prep(bs[1][0], bs[1][1], bs[1][2], blob=blob2, state=st2, seed=1, step=2)
            out = orig(pb, sd, seed_dev=seed_dev, adam_hyper=adam_hyper)
            if mode == "third":
                main.wait_stream(third)
                ops.step_state(m._tail_word(), 0, 1, 1e-3)
            return out
        m._step_body = body
    for i in range(30):
        m.train_step_graph(*bs[i % 8])
    g = m._graphs[(B, S, "adam", True)]
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(300):
            if mode == "base":
                m.train_step_graph(*bs[i % 8])
            else:
                g["graph"].replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 300)
    print(f"{mode}: {best * 1e6:.1f} us/step", flush=True)


for mode in ("base", "noprep", "third"):
    run(mode)
