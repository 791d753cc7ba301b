"""The D = 64 step at B = 512 with a fraction f of full-length (four-tile) sequences among Beauty-shaped ones: tile kernel against the fused
workgroup-per-item kernel (ms per step) -- where the plan's rule between the two should sit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
def mixed(c, n, seed, f):
    rng = np.random.default_rng(seed)
    out = []
    for b in bench.synth_batches(c, n, seed=seed):
        seq, pos, neg = (a.copy() for a in b)
        k = int(round(f * c["B"]))
        seq[:k, 1:] = rng.integers(1, c["items"] + 1, (k, c["S"] - 1)); seq[:k, 0] = 0
        pos[:k] = np.where(seq[:k] > 0, rng.integers(0, c["items"], seq[:k].shape), 0)
        neg[:k] = np.where(seq[:k] > 0, rng.integers(0, c["items"], seq[:k].shape), 0)
        out.append((seq, pos, neg))
    return out
for f in (0.0, 0.1, 0.2, 0.35, 0.5, 0.75):
    for tile in (True, False):
        m = SASRecEngine(cfg["items"], cfg["S"], cfg["D"], cfg["L"], dropout_rate=cfg["p_drop"], loss="BCE", lr=cfg["lr"], weight_decay=cfg["wd"], seed=1)
        m.tile_step = tile
        bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in mixed(cfg, 4, 11, f)]
        for i in range(6):
            m.train_step_graph(*bs[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            m.train_step_graph(*bs[i % 4])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 40 * 1e3
        m.check_handover()
        hdr = m.prepare_batch(*bs[0]).plan.view(torch.int32)[:8].cpu().numpy()
        print(f"f {f:4.2f} tile_step {tile!s:5s} mode {int(hdr[7])} tiles {int(hdr[1]):5d} long items {int(hdr[2]):4d}  {ms:.4f} ms/step", flush=True)
        del m
