"""One batch size, the next batch handed to every step (its preparation rides in the step's tail launch), 40 captured steps -- for a kernel
trace: bash scripts/prof_any_py.sh lb4096 40 scripts/lb_one_next.py 4096"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
big = dict(cfg, B=B)
m = SASRecEngine(cfg["items"], cfg["S"], cfg["D"], cfg["L"], dropout_rate=cfg["p_drop"], loss="BCE", lr=cfg["lr"], weight_decay=cfg["wd"], seed=1)
bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in bench.synth_batches(big, 4, seed=11)]
for i in range(48):
    m.train_step_graph(*bs[i % 4], next_batch=bs[(i + 1) % 4])
torch.cuda.synchronize()
m.check_handover()
