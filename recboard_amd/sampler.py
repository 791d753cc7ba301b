"""Device-side batch assembly: the reference's training datapipes as ONE engine launch per batch.

    reference chain (SASRec/main.py:143-157)                                     here
    shuffled_seqs_source(maxlen) -> seq_train_yielding_pos_(1, -1)               DeviceSeqSampler: a device permutation per epoch,
      -> seq_train_sampling_neg_(1) -> add_(1, (ISeq,)) -> lpad_ -> batch_         re_seq_train_sample per batch (csrc/sampler.hip)
    choiced_user_ids_source -> gen_train_sampling_pos_ -> ..._neg_(1)            DeviceGenSampler: re_gen_train_sample per batch
      (MF-BPR/main.py:60-68)

The training interactions are uploaded once (CSR over users: chronological items + the same items sorted, the "seen" probe); a batch is
device tensors, so the fused step reads it without a host-to-device copy.  `device_pipe(pipe)` is what `freerec`'s chained pipes
(freerec/data/postprocessing.py) hand over to when they are asked to run on a device (`pipe.to_(device)`)."""
import numpy as np
import torch

from . import lib
from .ops import _p, _stream


class DeviceInteractions:
    """ptr / items (chronological) / sorted_items of the training split, on the device."""

    def __init__(self, ptr, items, num_items, device="cuda"):
        ptr, items = np.asarray(ptr, np.int64), np.asarray(items, np.int64)
        srt = items.copy()
        for u in np.nonzero(np.diff(ptr) > 1)[0]:
            srt[ptr[u]:ptr[u + 1]].sort()
        self.device = torch.device(device)
        self.ptr, self.items, self.sorted = (torch.from_numpy(a).to(self.device) for a in (ptr, items, srt))
        self.num_users, self.num_items = len(ptr) - 1, int(num_items)
        lens = np.diff(ptr)
        self.users_ge2 = torch.from_numpy(np.nonzero(lens >= 2)[0]).to(self.device)
        self.users_ge1 = torch.from_numpy(np.nonzero(lens >= 1)[0]).to(self.device)
        self.nnz = int(ptr[-1])

    @classmethod
    def from_dataset(cls, ds, device="cuda"):
        """A freerec-surface RecDataSet (its training split) or anything with `train_seq(u)` / num_users / num_items (recboard_amd.data)."""
        if hasattr(ds, "_seqs"):
            ptr, items = ds._seqs["train"]
            return cls(ptr, items, ds.num_items, device)
        seqs = [np.asarray(ds.train_seq(u), np.int64) for u in range(ds.num_users)]
        ptr = np.zeros(len(seqs) + 1, np.int64)
        np.cumsum([len(s) for s in seqs], out=ptr[1:])
        return cls(ptr, np.concatenate(seqs) if seqs else np.zeros(0, np.int64), ds.num_items, device)


def seq_train_sample(inter, order, b0, B, S, seed, step, out=None):
    """One batch of the SASRec training chain (re_seq_train_sample).  -> (users [B], seq, pos, neg [B, S]) int64 on the device."""
    dev = inter.device
    if out is None:
        out = (torch.empty(B, dtype=torch.int64, device=dev),) + tuple(torch.empty((B, S), dtype=torch.int64, device=dev) for _ in range(3))
    users, seq, pos, neg = out
    lib.check(lib.load().re_seq_train_sample(_p(inter.ptr), _p(inter.items), _p(inter.sorted), _p(order), order.numel(), int(b0), B, S,
                                             inter.num_items, int(seed) & 0xFFFFFFFF, int(step) & 0xFFFFFFFF, _p(users), _p(seq), _p(pos),
                                             _p(neg), _stream()), "re_seq_train_sample")
    return users, seq, pos, neg


def gen_train_sample(inter, B, seed, step):
    """One batch of (user, positive, unseen negative) triplets (re_gen_train_sample).  -> three [B, 1] int64 device tensors."""
    dev = inter.device
    users, pos, neg = (torch.empty((B, 1), dtype=torch.int64, device=dev) for _ in range(3))
    order = inter.users_ge1
    lib.check(lib.load().re_gen_train_sample(_p(inter.ptr), _p(inter.items), _p(inter.sorted), _p(order), order.numel(), B, inter.num_items,
                                             int(seed) & 0xFFFFFFFF, int(step) & 0xFFFFFFFF, _p(users), _p(pos), _p(neg), _stream()),
              "re_gen_train_sample")
    return users, pos, neg


class SampleTicket:
    """What a fused sampler hands out instead of tensors: the batch is rows b0 .. b0 + B of `order`, to be sampled by the step's own
    preparation launch (ops.sasrec_sample_prep) with (seed, step)."""
    __slots__ = ("inter", "order", "b0", "B", "S", "seed", "step", "users")

    def __len__(self):
        return self.B


class DeviceSeqSampler:
    """An epoch = every user with >= 2 training items once, in a fresh device permutation; batches of {User, ISeq, IPos, INeg}.
    fused=True: batches of {"Sample": SampleTicket} -- the engine's preparation launch samples the rows itself (the same rows and draws)."""

    def __init__(self, inter, maxlen, batch_size, seed=1, keys=("User", "ISeq", "IPos", "INeg"), fused=False):
        self.inter, self.S, self.B, self.seed, self.keys = inter, int(maxlen), int(batch_size), int(seed), keys
        self.gen = torch.Generator(device=inter.device).manual_seed(seed)
        self.step, self.fused = 0, bool(fused)

    def __len__(self):
        return (self.inter.users_ge2.numel() + self.B - 1) // self.B

    def __iter__(self):
        us = self.inter.users_ge2
        order = us[torch.randperm(us.numel(), device=us.device, generator=self.gen)]
        for b0 in range(0, us.numel(), self.B):
            B = min(self.B, us.numel() - b0)
            self.step += 1
            if self.fused:
                t = SampleTicket()
                t.inter, t.order, t.b0, t.B, t.S, t.seed, t.step, t.users = self.inter, order, b0, B, self.S, self.seed, self.step, None
                yield {"Sample": t}
                continue
            users, seq, pos, neg = seq_train_sample(self.inter, order, b0, B, self.S, self.seed, self.step)
            yield dict(zip(self.keys, (users, seq, pos, neg)))


class DeviceGenSampler:
    """An epoch = one triplet per training interaction (MF-BPR/main.py:60-68), `batch_size` per batch."""

    def __init__(self, inter, batch_size, seed=1, keys=("User", "IPos", "INeg")):
        self.inter, self.B, self.seed, self.keys, self.step = inter, int(batch_size), int(seed), keys, 0

    def __len__(self):
        return (self.inter.nnz + self.B - 1) // self.B

    def __iter__(self):
        for _ in range(len(self)):
            self.step += 1
            yield dict(zip(self.keys, gen_train_sample(self.inter, self.B, self.seed, self.step)))


def device_pipe(pipe, fused=False):
    """A recorded freerec pipe (freerec/data/postprocessing.py) -> a device sampler yielding the same {Field: tensor} batches, when the
    chain is one the engine samples on the device; None otherwise (the pipe then runs its vectorised host path).
    fused (SASRec chain only): {"Sample": ticket} batches for SASRecEngine.train_step_graph_sampled."""
    from freerec.data import tags as T
    from freerec.data.postprocessing import _item_roles
    ds = pipe.ds
    if not hasattr(ds, "_seqs") or pipe.batch_size is None:
        return None
    R = _item_roles(ds)
    names = [n for n, _ in pipe.ops]
    inter = getattr(ds, "_device_inter", None)
    if pipe.source == "shuffled_seqs" and names == ["seq_pos", "seq_neg", "add", "lpad"]:
        (_, a), (_, b), (_, c), (_, d) = pipe.ops
        ok = (a == dict(start=1, end=-1) and b["k"] == 1 and c["offset"] == 1 and tuple(c["fields"]) == (R["ISeq"],) and d["value"] == 0
              and set(d["fields"]) == {R["ISeq"], R["IPos"], R["INeg"]} and pipe.kw.get("maxlen") in (None, d["maxlen"]))
        if not ok:
            return None
        if inter is None or inter.device != pipe.device:
            inter = ds._device_inter = DeviceInteractions.from_dataset(ds, pipe.device)
        smp = getattr(pipe, "_device_sampler", None)
        if smp is None:
            smp = pipe._device_sampler = DeviceSeqSampler(inter, d["maxlen"], pipe.batch_size, seed=int(pipe.rng.integers(1 << 31)),
                                                          keys=(R["User"], R["ISeq"], R["IPos"], R["INeg"]), fused=fused)

        def batches():
            for bt in smp:
                bt[R["Size"]] = len(bt["Sample"]) if "Sample" in bt else int(bt[R["User"]].numel())
                yield bt
        return batches()
    if pipe.source == "choiced_user_ids" and names == ["gen_pos", "gen_neg"] and pipe.ops[1][1]["k"] == 1:
        if inter is None or inter.device != pipe.device:
            inter = ds._device_inter = DeviceInteractions.from_dataset(ds, pipe.device)
        smp = getattr(pipe, "_device_sampler", None)
        if smp is None:
            smp = pipe._device_sampler = DeviceGenSampler(inter, pipe.batch_size, seed=int(pipe.rng.integers(1 << 31)),
                                                          keys=(R["User"], R["IPos"], R["INeg"]))

        def batches():
            for bt in smp:
                bt[R["Size"]] = pipe.batch_size
                yield bt
        return batches()
    return None
