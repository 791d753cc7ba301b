"""Where the host time of Coach.train_per_epoch over pinned HOST batches goes (cProfile; the GPU is not the limit there)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from recboard_amd.coach import Coach
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
hb = bench.synth_batches(cfg, 10, seed=77)
nb = 200
pipe = [{"User": torch.arange(cfg["B"]), "ISeq": torch.from_numpy(hb[i % 10][0]).pin_memory(), "IPos": torch.from_numpy(hb[i % 10][1]).pin_memory(),
         "INeg": torch.from_numpy(hb[i % 10][2]).pin_memory()} for i in range(nb)]
if os.environ.get("DEV_PIPE"):        # the same batches already on the device: what the loop costs without the copies
    pipe = [{k: (v.cuda() if k != "User" else v) for k, v in d.items()} for d in pipe]
coach = Coach(m, pipe, monitors=["LOSS"], kind="seq")
coach.train_per_epoch(0)
torch.cuda.synchronize()
import time
for rep in range(3):
    for packed in (True, False):
        coach.pack_copies = packed
        coach.train_per_epoch(1); torch.cuda.synchronize()
        t0 = time.perf_counter(); coach.train_per_epoch(1); torch.cuda.synchronize()
        print("epoch (one packed copy per batch: %s): %.1f us per step" % (packed, (time.perf_counter() - t0) / nb * 1e6), flush=True)
pr = cProfile.Profile(); pr.enable(); coach.train_per_epoch(2); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
