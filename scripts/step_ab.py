"""A/B of the captured SASRec/Beauty step with an engine attribute toggled:  python scripts/step_ab.py fork_wgrad [more attributes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import lib
if os.environ.get("RECENGINE_LIB"):          # (an experimental build of the library: make a variant .so next to librecengine.so)
    lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), os.environ["RECENGINE_LIB"])
import bench
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
hb = bench.synth_batches(cfg, 8, 1)
bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in hb]
for attr in [None] + sys.argv[1:]:
    for val in ((True,) if attr is None else (False, True)):
        m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
        if attr:
            setattr(m, attr, val)
        for i in range(30):
            m.train_step_graph(*bs[i % 8], next_batch=bs[(i + 1) % 8] if (m.pipelined_prep or os.environ.get('STEP_AB_NEXT')) else None)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for i in range(300):
                loss = m.train_step_graph(*bs[i % 8], next_batch=bs[(i + 1) % 8] if (m.pipelined_prep or os.environ.get('STEP_AB_NEXT')) else None)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 300)
        print(f"{attr}={val}: {best * 1e6:.1f} us/step  loss {float(loss):.5f}", flush=True)
