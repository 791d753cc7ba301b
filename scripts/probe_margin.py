"""The adoption probe's margin: per-tensor distance of the engine's gradients from the script's own double-precision step for the genuine
SASRec script under random initialisations (what the tolerance must let through) and for a look-alike whose attention reads layer-normed
keys and values (what it must stop).    RECENGINE_PROBE_REPORT=1 python scripts/probe_margin.py"""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["RECENGINE_PROBE_REPORT"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402
from test_freerec_compat import import_script, toy_dataset  # noqa: E402

from recboard_amd import bridge  # noqa: E402
bridge.GRAD_TOL = bridge.GRAD_TOL_SMOOTH = bridge.GRAD_TOL_UPSTREAM = 1e9          # (report only)
mod = import_script(os.path.join(ROOT, "examples", "SASRec", "main.py"), "_probe_margin", ["--dropout-rate", "0.5", "--loss", "BCE"])


class NormedKV(mod.SASRec):
    def encode(self, data):
        seq = data[self.ISeq]
        pad = (seq == self.PADDING_VALUE).unsqueeze(-1)
        x = self.Item.embeddings(seq) * (mod.cfg.embedding_dim ** 0.5) + self.Position(self.positions)
        x = self.embdDropout(x).masked_fill(pad, 0.0)
        for l in range(self.num_blocks):
            q = self.attnLNs[l](x)
            x = self.attnLayers[l](q, q, q, attn_mask=self.attnMask, need_weights=False)[0] + x
            x = self.fwdLayers[l](self.fwdLNs[l](x)).masked_fill(pad, 0.0)
        return self.lastLN(x), self.Item.embeddings.weight[self.NUM_PADS:]


N, B, S = 12101, 512, 50
rng = np.random.default_rng(3)
ds = toy_dataset(B, N)
lens = np.clip(rng.geometric(1 / 5.9, B) + 1, 1, S - 1)
seq = np.zeros((B, S), np.int64)
for b in range(B):
    seq[b, S - lens[b]:] = rng.integers(1, N + 1, lens[b])
pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
cfg = mod.cfg
cfg.device, cfg.engine, cfg.epochs, cfg.eval_freq, cfg.monitors, cfg.which4best = "cuda:0", "auto", 1, 1, ["LOSS"], "LOSS"
for cls, reps in ((mod.SASRec, int(sys.argv[1]) if len(sys.argv) > 1 else 6), (NormedKV, 2)):
    for r in range(reps):
        torch.manual_seed(100 + r)
        model = cls(ds)
        batch = {model.User: torch.arange(B), model.ISeq: torch.from_numpy(seq), model.IPos: torch.from_numpy(pos), model.INeg: torch.from_numpy(neg), model.Size: B}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mod.CoachForSASRec(dataset=ds, trainpipe=[batch], validpipe=None, testpipe=None, model=model, cfg=cfg)
