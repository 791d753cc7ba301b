import sys; sys.path.insert(0, "/root/repo")
import torch, bench
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
for _ in range(3):
    r = bench.sampler_rates(cfg, m)
    print({k: v for k, v in r.items() if k != "what"}, flush=True)
