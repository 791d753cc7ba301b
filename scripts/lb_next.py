"""Large batches with and without the NEXT batch handed to the step (train_step_graph(next_batch=...): its preparation rides in this step's tail
launch or on a side stream instead of standing in front of the next step): python scripts/lb_next.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
for B in (1024, 2048, 4096, 8192):
    for nxt in (False, True):
        big = dict(cfg, B=B)
        m = SASRecEngine(cfg["items"], cfg["S"], cfg["D"], cfg["L"], dropout_rate=cfg["p_drop"], loss="BCE", lr=cfg["lr"], weight_decay=cfg["wd"], seed=1)
        bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in bench.synth_batches(big, 4, seed=11)]
        step = (lambda i: m.train_step_graph(*bs[i % 4], next_batch=bs[(i + 1) % 4])) if nxt else (lambda i: m.train_step_graph(*bs[i % 4]))
        for i in range(8):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            loss = step(i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 40 * 1e3
        m.check_handover()
        print(f"B {B:5d} next_batch {nxt!s:5s}  {ms:.4f} ms/step  {B / ms / 1e3:.2f} M samples/s  loss {float(loss):.6f}", flush=True)
        del m
