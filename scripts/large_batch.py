"""The D = 64 step at larger batches, tile kernel against the workgroup-per-item kernels (ms per step, samples/s)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
import numpy as np
def full_batches(c, n, seed):      # every sequence S - 1 items long: four tiles each
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        seq = rng.integers(1, c["items"] + 1, (c["B"], c["S"])); seq[:, 0] = 0
        out.append((seq, np.where(seq > 0, rng.integers(0, c["items"], seq.shape), 0), np.where(seq > 0, rng.integers(0, c["items"], seq.shape), 0)))
    return out
for B in (-512, 1024, 2048, 4096, 8192):
    for tile in (True, False):        # (False: tile_step off, the FUSED workgroup-per-item kernel enc_step_k -- what the plan chose before)
        full = B < 0
        B = abs(B)
        big = dict(cfg, B=B)
        m = SASRecEngine(cfg["items"], cfg["S"], cfg["D"], cfg["L"], dropout_rate=cfg["p_drop"], loss="BCE", lr=cfg["lr"], weight_decay=cfg["wd"], seed=1)
        m.tile_step = tile
        bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in (full_batches(big, 4, 11) if full else bench.synth_batches(big, 4, seed=11))]
        for i in range(6):
            m.train_step_graph(*bs[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(30):
            m.train_step_graph(*bs[i % 4])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 30 * 1e3
        m.check_handover()
        hdr = m.prepare_batch(*bs[0]).plan.view(torch.int32)[:8].cpu().numpy()
        print(f"B {B:5d} tile_step {tile!s:5s} mode {int(hdr[7])} tiles {int(hdr[1]):5d} long items {int(hdr[2]):4d}  {ms:.4f} ms/step  {B / ms / 1e3:.2f} M samples/s", flush=True)
        del m
        if full:
            B = -B
