"""Host cost of Coach.train_per_epoch over HOST batches (the bench's coach_loop leg): cProfile of one epoch of 300 steps + the wall time
with and without waiting for the GPU.    python scripts/coach_host_profile.py"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from recboard_amd.coach import Coach  # noqa: E402
from recboard_amd.sasrec import SASRecEngine  # noqa: E402
cfg = bench.BEAUTY
nb = 300
hb = bench.synth_batches(cfg, 10, seed=77)
pipe = [{"User": torch.arange(cfg["B"]), "ISeq": torch.from_numpy(hb[i % 10][0]).pin_memory(),
         "IPos": torch.from_numpy(hb[i % 10][1]).pin_memory(), "INeg": torch.from_numpy(hb[i % 10][2]).pin_memory()} for i in range(nb)]
model = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
model.train()
coach = Coach(model, pipe, monitors=["LOSS"], kind="seq")
coach.train_per_epoch(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
coach.train_per_epoch(1)
torch.cuda.synchronize()
print("epoch: %.4f ms per step" % ((time.perf_counter() - t0) / nb * 1e3))
pr = cProfile.Profile()
pr.enable()
coach.train_per_epoch(2)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
