"""Vectorised batch assembly for the engine: the row contract of the reference's datapipes without their per-row Python
generators.

Reference pipes restated (freerec datapipes are external; row semantics evidenced by HSTU/sampler.py:47-125):
  SASRec train  `shuffled_seqs_source(maxlen) -> seq_train_yielding_pos_(1, -1) -> seq_train_sampling_neg_(1)
                 -> add_(NUM_PADS, (ISeq,)) -> lpad_(maxlen, (ISeq, IPos, INeg), 0) -> batch_ -> tensor_`  (SASRec/main.py:143-157)
       one row per user: ISeq = seq[:-1] (+1, left-padded with 0), IPos = seq[1:] (0-based, 0 on pads),
       INeg = one uniform negative per position, not in the user's seen set.
  MF/LightGCN   `choiced_user_ids_source -> gen_train_sampling_pos_ -> gen_train_sampling_neg_(1)`  (MF-BPR/main.py:60-68)
       one row per draw: a uniformly chosen user, one of its positives, one unseen negative; an "epoch" = #train interactions.
  eval (full)   one row per user: {User, ISeq (train [+valid] history), IUnseen = (target,), ISeen = history}  (HSTU/sampler.py:117-125)

Everything is numpy on the host (batch assembly is not on the GPU hot path); tensors are created per batch.
"""
import numpy as np
import torch


class SyntheticSeqDataset:
    """Leave-one-out next-item dataset with planted first-order structure: with probability `p_follow` the next item is
    `perm[prev]`, otherwise Zipf-random.  Shapes follow the Beauty statistics when called with its cardinalities."""

    def __init__(self, num_users, num_items, mean_len=8.9, min_len=5, max_len=200, p_follow=0.7, seed=1):
        rng = np.random.default_rng(seed)
        self.num_users, self.num_items = num_users, num_items
        perm = rng.permutation(num_items)
        w = 1.0 / np.arange(1, num_items + 1)
        w /= w.sum()
        lens = np.clip(rng.geometric(1.0 / max(mean_len - min_len + 1, 1.0), num_users) + min_len - 1, min_len, max_len)
        self.seqs = []
        for L in lens:
            s = np.empty(L, np.int64)
            s[0] = rng.choice(num_items, p=w)
            follow = rng.random(L) < p_follow
            rnd = rng.choice(num_items, L, p=w)
            for t in range(1, L):
                s[t] = perm[s[t - 1]] if follow[t] else rnd[t]
            self.seqs.append(s)
        self.perm = perm

    def train_seq(self, u):
        return self.seqs[u][:-2]

    def valid_target(self, u):
        return self.seqs[u][-2]

    def test_target(self, u):
        return self.seqs[u][-1]

    def num_train_interactions(self):
        return int(sum(len(s) - 2 for s in self.seqs))


class ExplicitSeqDataset(SyntheticSeqDataset):
    """The same leave-one-out interface over GIVEN per-user item sequences (last item = test target, the one before = valid target)."""

    def __init__(self, seqs, num_items):
        self.seqs = [np.asarray(s, np.int64) for s in seqs]
        self.num_users, self.num_items, self.perm = len(self.seqs), num_items, None


def _lpad(rows, maxlen, offset=0):
    out = np.zeros((len(rows), maxlen), np.int64)
    for i, r in enumerate(rows):
        r = r[-maxlen:]
        if len(r):
            out[i, maxlen - len(r):] = r + offset
    return out


class SeqTrainSampler:
    """SASRec/main.py:143-157, vectorised: padded ISeq/IPos built once; per epoch a shuffle + one vectorised negative draw
    with rejection against the user's seen set."""

    def __init__(self, dataset, maxlen, batch_size, seed=1):
        self.ds, self.S, self.B = dataset, maxlen, batch_size
        self.rng = np.random.default_rng(seed)
        users = [u for u in range(dataset.num_users) if len(dataset.train_seq(u)) >= 2]
        self.users = np.asarray(users)
        full = [dataset.train_seq(u) for u in users]
        tr = [s[-maxlen:] for s in full]          # shuffled_seqs_source(maxlen): the sequence is cut BEFORE the target is split off (HSTU/sampler.py:28-31)
        self.iseq = _lpad([s[:-1] for s in tr], maxlen, offset=1)      # NUM_PADS offset on ISeq only
        self.ipos = _lpad([s[1:] for s in tr], maxlen)
        self.mask = self.iseq != 0
        n = dataset.num_items
        self.seen = np.zeros((len(users), n), bool) if len(users) * n <= 4e8 else None
        if self.seen is not None:
            for i, s in enumerate(full):
                self.seen[i, s] = True

    def _negatives(self, rows):
        neg = self.rng.integers(0, self.ds.num_items, (len(rows), self.S))
        if self.seen is not None:
            for _ in range(8):
                bad = self.seen[rows[:, None], neg] & self.mask[rows]
                if not bad.any():
                    break
                neg[bad] = self.rng.integers(0, self.ds.num_items, int(bad.sum()))
        return np.where(self.mask[rows], neg, 0)

    def __len__(self):
        return (len(self.users) + self.B - 1) // self.B

    def __iter__(self):
        order = self.rng.permutation(len(self.users))
        for i in range(0, len(order), self.B):
            rows = order[i:i + self.B]
            yield {"User": torch.from_numpy(self.users[rows]), "ISeq": torch.from_numpy(self.iseq[rows]),
                   "IPos": torch.from_numpy(self.ipos[rows]), "INeg": torch.from_numpy(self._negatives(rows))}


class GenTrainSampler:
    """MF-BPR/main.py:60-68, vectorised: `steps_per_epoch * B` (user, positive, unseen negative) triplets per epoch."""

    def __init__(self, dataset, batch_size, seed=1):
        self.ds, self.B = dataset, batch_size
        self.rng = np.random.default_rng(seed)
        self.tr = [dataset.train_seq(u) for u in range(dataset.num_users)]
        self.lens = np.asarray([len(s) for s in self.tr])
        self.flat = np.concatenate(self.tr)
        self.ptr = np.concatenate([[0], np.cumsum(self.lens)])
        self.users = np.nonzero(self.lens > 0)[0]
        self.seen = [set(s.tolist()) for s in self.tr]

    def __len__(self):
        return (int(self.lens.sum()) + self.B - 1) // self.B

    def __iter__(self):
        for _ in range(len(self)):
            u = self.rng.choice(self.users, self.B)
            pos = self.flat[self.ptr[u] + (self.rng.random(self.B) * self.lens[u]).astype(np.int64)]
            neg = self.rng.integers(0, self.ds.num_items, self.B)
            for i in range(self.B):
                while int(neg[i]) in self.seen[u[i]]:
                    neg[i] = self.rng.integers(0, self.ds.num_items)
            yield {"User": torch.from_numpy(u).unsqueeze(1), "IPos": torch.from_numpy(pos).unsqueeze(1),
                   "INeg": torch.from_numpy(neg).unsqueeze(1)}


class EvalSampler:
    """Full-ranking rows (HSTU/sampler.py:117-125): history, one held-out target, seen = history."""

    def __init__(self, dataset, maxlen, batch_size, mode="valid"):
        self.ds, self.S, self.B, self.mode = dataset, maxlen, batch_size, mode

    def __iter__(self):
        ds = self.ds
        for i in range(0, ds.num_users, self.B):
            us = np.arange(i, min(i + self.B, ds.num_users))
            hist = [ds.train_seq(u) if self.mode == "valid" else ds.seqs[u][:-1] for u in us]
            tgt = [ds.valid_target(u) if self.mode == "valid" else ds.test_target(u) for u in us]
            yield {"User": torch.from_numpy(us), "ISeq": torch.from_numpy(_lpad(hist, self.S, offset=1)),
                   "IUnseen": [[int(t)] for t in tgt], "ISeen": [np.unique(h).tolist() for h in hist], "Size": len(us)}
