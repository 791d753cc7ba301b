"""GPU time of the batch-preparation launch in its three forms (plain, + weights, sample + prepare) at the bench shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from recboard_amd import ops
from recboard_amd.sampler import DeviceInteractions, seq_train_sample
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
B, S, N, U = 512, 50, cfg["items"], cfg["users"]
m = SASRecEngine(N, S, 64, 2, dropout_rate=0.5)
seq, pos, neg = (torch.from_numpy(a).cuda() for a in bench.synth_batches(cfg, 1, 1)[0])
blob = torch.zeros(ops.prep_layout(B, S)[1], dtype=torch.uint8, device="cuda")
state = torch.zeros(4, dtype=torch.int32, device="cuda")
rng = np.random.default_rng(5)
lens = np.clip(rng.geometric(1 / 5.9, U) + 2, 2, 200)
ptr = np.zeros(U + 1, np.int64); np.cumsum(lens, out=ptr[1:])
w = 1.0 / np.arange(1, N + 1); w /= w.sum()
inter = DeviceInteractions(ptr, rng.choice(N, int(ptr[-1]), p=w), N)
order = inter.users_ge2[torch.randperm(inter.users_ge2.numel(), device="cuda")]
wts = m._prep_weights(B, S)
print("plain        %.1f us" % (1e3 * bench.graph_time_ms(lambda: ops.sasrec_batch_prep(seq, pos, neg, blob=blob, state=state))))
print("+ weights    %.1f us" % (1e3 * bench.graph_time_ms(lambda: ops.sasrec_batch_prep(seq, pos, neg, blob=blob, state=state, weights=wts))))
print("sampler      %.1f us" % (1e3 * bench.graph_time_ms(lambda: seq_train_sample(inter, order, 0, B, S, 3, 1))))
print("sample+prep  %.1f us" % (1e3 * bench.graph_time_ms(lambda: ops.sasrec_sample_prep(inter, order, 0, B, S, 3, 1, blob, state=state, weights=wts))))
