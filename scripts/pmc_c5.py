"""Config-5 steps (SASRec d = 128 on the 100 M-item table) launched eagerly over DISTINCT batches: the target of the FETCH_SIZE / WRITE_SIZE
`rocprofv3 --pmc` passes behind bench.py's `config5.hbm_traffic_per_step` (scripts/make_profiles.py sums the engine's kernels per step).
usage: python scripts/pmc_c5.py [steps, default 20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench_legs
from recboard_amd.large import SASRecLargeTableEngine
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, D, B, S = int(os.environ.get("RECBENCH_C5_ITEMS", 100_000_000)), 128, 512, 50
eng = SASRecLargeTableEngine(N, S, D, 2, dropout_rate=0.5, loss="BCE", lr=1e-3, weight_decay=1e-6, seed=1)
bs = bench_legs.c5_batches(np.random.default_rng(1), steps, N, B, S)
for b in bs:
    eng.train_step(*b)           # eager launches: the kernels the captured step replays, one counter record each
torch.cuda.synchronize()
print("steps", steps)
