"""The chained datapipe surface (`dataset.train().shuffled_seqs_source(maxlen).seq_train_yielding_pos_(...).seq_train_sampling_neg_(...)
.add_(...).lpad_(...).batch_(B).tensor_()`, SASRec/main.py:143-157; `choiced_user_ids_source().gen_train_sampling_pos_()
.gen_train_sampling_neg_(...)`, MF-BPR/main.py:60-68; `valid_sampling_ / test_sampling_`; row contracts evidenced by
HSTU/sampler.py:47-125) -- VECTORISED: a chain is recorded, and iterating it draws an epoch's row seeds, cuts them into batches and
applies every op to a whole batch with NumPy (no per-row Python generators, no worker processes).  With `device=` the SASRec
training chain is sampled on the GPU instead (recboard_amd.sampler: SURVEY.md section 8f-1).

Batch = {Field: tensor | list}: keys are the model's fields (equal by name + tags).  Row contracts:
  seq train   ISeq = seq[:end_idx_for_input], IPos = seq[start_idx_for_target:], INeg = uniform items not in the user's training
              set, one (or K) per position; `add_(offset, (ISeq,))` shifts ids past the padding value; `lpad_` left-pads to maxlen.
  gen train   User drawn uniformly (with replacement, one per training interaction per epoch) among users with items,
              IPos one of its training items, INeg K uniform unseen items.
  valid/test  per user: ISeq = history (train; + valid for test), IUnseen = the held-out items, ISeen = history (ragged lists)."""
import numpy as np
import torch

from . import tags as T
from .fields import Field


def _item_roles(ds):
    item, user = ds.fields[T.ITEM, T.ID], ds.fields[T.USER, T.ID]
    return dict(User=user, Item=item, ISeq=item.fork(T.SEQUENCE), IPos=item.fork(T.POSITIVE), INeg=item.fork(T.NEGATIVE),
                IUnseen=item.fork(T.UNSEEN), ISeen=item.fork(T.SEEN), Size=Field("SIZE", T.SIZE))


class Pipe:
    def __init__(self, split, source, kw):
        self.split, self.ds, self.source, self.kw = split, split.ds, source, kw
        self.ops = []
        self.batch_size, self.as_tensor = None, False
        self.rng = np.random.default_rng(1)
        self.device = None

    # ---- recording
    def _op(self, name, **kw):
        self.ops.append((name, kw))
        return self

    def seq_train_yielding_pos_(self, start_idx_for_target=1, end_idx_for_input=-1):
        return self._op("seq_pos", start=start_idx_for_target, end=end_idx_for_input)

    def seq_train_sampling_neg_(self, num_negatives=1, unseen_only=True):
        return self._op("seq_neg", k=num_negatives)

    def gen_train_sampling_pos_(self):
        return self._op("gen_pos")

    def gen_train_sampling_neg_(self, num_negatives=1, unseen_only=True):
        return self._op("gen_neg", k=num_negatives)

    def valid_sampling_(self, ranking="full", num_negatives=100):
        return self._op("eval", mode="valid", ranking=ranking, k=num_negatives)

    def test_sampling_(self, ranking="full", num_negatives=100):
        return self._op("eval", mode="test", ranking=ranking, k=num_negatives)

    def add_(self, offset=1, modified_fields=()):
        return self._op("add", offset=offset, fields=tuple(modified_fields))

    def lprune_(self, maxlen, modified_fields=()):
        return self._op("lprune", maxlen=maxlen, fields=tuple(modified_fields))

    def rprune_(self, maxlen, modified_fields=()):
        return self._op("rprune", maxlen=maxlen, fields=tuple(modified_fields))

    def lpad_(self, maxlen, modified_fields=(), padding_value=0):
        return self._op("lpad", maxlen=maxlen, fields=tuple(modified_fields), value=padding_value)

    def rpad_(self, maxlen, modified_fields=(), padding_value=0):
        return self._op("rpad", maxlen=maxlen, fields=tuple(modified_fields), value=padding_value)

    def batch_(self, batch_size, drop_last=False):
        self.batch_size = int(batch_size)
        return self

    def tensor_(self):
        self.as_tensor = True
        return self

    def shard_(self):
        return self

    def seed_(self, seed):
        self.rng = np.random.default_rng(seed)
        return self

    def to_(self, device, fused=False):
        """recengine extension: sample the epoch on `device` where a device sampler exists for the chain (SASRec training).
        fused: hand out sample tickets instead of tensors -- the engine's batch-preparation launch samples the rows itself."""
        self.device, self.fused = torch.device(device), bool(fused)
        return self

    # ---- an epoch's row seeds
    def _seeds(self):
        ds = self.ds
        ptr, _ = ds._seqs["train"] if hasattr(ds, "_seqs") else (None, None)
        if self.source in ("shuffled_seqs", "ordered_seqs"):
            users = np.nonzero(np.diff(ptr) >= 2)[0]            # a training row needs an input and a target
            return self.rng.permutation(users) if self.source == "shuffled_seqs" else users
        if self.source == "choiced_user_ids":
            users = np.nonzero(np.diff(ptr) > 0)[0]
            return self.rng.choice(users, ds.trainsize)
        if self.source == "ordered_user_ids":
            ptr_e, _ = ds._seqs[self._eval_mode() or self.split.mode]
            return np.nonzero(np.diff(ptr_e) > 0)[0] if self._eval_mode() else np.arange(ds.num_users)
        n = self.split.datasize
        return self.rng.permutation(n) if self.source.startswith("shuffled") else np.arange(n)

    def _eval_mode(self):
        for name, kw in self.ops:
            if name == "eval":
                return kw["mode"]
        return None

    def __len__(self):
        n = len(self._seeds())
        return (n + self.batch_size - 1) // self.batch_size if self.batch_size else n

    # ---- vectorised pieces
    def _unseen(self, users, shape):
        """Uniform items not in the users' training sets: draw, test (user, item) against the sorted key array, redraw the hits."""
        ds = self.ds
        neg = self.rng.integers(0, ds.num_items, shape)
        u = np.broadcast_to(users.reshape((-1,) + (1,) * (len(shape) - 1)), shape)
        for _ in range(64):
            key = u * ds.num_items + neg
            pos = np.searchsorted(ds._seen_keys, key)
            bad = (pos < len(ds._seen_keys)) & (ds._seen_keys[np.minimum(pos, len(ds._seen_keys) - 1)] == key)
            if not bad.any():
                break
            neg = np.where(bad, self.rng.integers(0, ds.num_items, shape), neg)
        return neg

    def _rows(self, seeds):
        """One batch of rows from its seeds: {Field: list of arrays | array}."""
        ds = self.ds
        R = _item_roles(ds) if hasattr(ds, "_seqs") else {}
        row = {}
        if self.source in ("shuffled_seqs", "ordered_seqs"):
            ml = self.kw.get("maxlen")
            seqs = [ds.seq("train", u) for u in seeds]
            row[R["User"]] = seeds
            row["_seq"] = [s[-ml:] if ml else s for s in seqs]           # to_seqs(maxlen): cut BEFORE the target is split off (HSTU/sampler.py:28-31)
        elif self.source in ("choiced_user_ids", "ordered_user_ids"):
            row[R["User"]] = seeds
        elif self.source == "shuffled_pairs":
            s = ds.splits[self.split.mode]
            row[R["User"]], row[R["IPos"]] = np.asarray(s["USER"], np.int64)[seeds], np.asarray(s["ITEM"], np.int64)[seeds]
        else:                                                            # interaction rows of a prediction dataset: every column
            s = ds.splits[self.split.mode]
            for f in ds.fields:
                if f.name in s:
                    col = np.asarray(s[f.name])[seeds]
                    row[f] = col.astype(np.int64) if f.match(T.SPARSE) or f.match(T.ID) or f.match(T.LABEL) else col.astype(np.float32)
        for name, kw in self.ops:
            if name == "seq_pos":
                sq = row.pop("_seq")
                end = kw["end"] if kw["end"] != 0 else None
                row[R["ISeq"]] = [s[:end] for s in sq]
                row[R["IPos"]] = [s[kw["start"]:] for s in sq]
            elif name == "seq_neg":
                lens = np.asarray([len(p) for p in row[R["IPos"]]])
                users = np.repeat(np.asarray(row[R["User"]]), lens)
                k = kw["k"]
                neg = self._unseen(users, (int(lens.sum()),) if k == 1 else (int(lens.sum()), k))
                row[R["INeg"]] = np.split(neg, np.cumsum(lens)[:-1]) if len(lens) else []
            elif name == "gen_pos":
                u = np.asarray(row[R["User"]])
                ptr, items = ds._seqs["train"]
                row[R["IPos"]] = items[ptr[u] + (self.rng.random(len(u)) * (ptr[u + 1] - ptr[u])).astype(np.int64)].reshape(-1, 1)
                row[R["User"]] = u.reshape(-1, 1)
            elif name == "gen_neg":
                u = np.asarray(row[R["User"]]).reshape(-1)
                row[R["INeg"]] = self._unseen(u, (len(u), kw["k"]))
            elif name == "eval":
                u = np.asarray(row[R["User"]])
                hist = [ds.seq("train", x) if kw["mode"] == "valid" else np.concatenate([ds.seq("train", x), ds.seq("valid", x)]) for x in u]
                row[R["ISeq"]] = hist
                row[R["IUnseen"]] = [ds.seq(kw["mode"], x) for x in u]
                row[R["ISeen"]] = [np.unique(h) for h in hist]
                # A NEXT-ITEM pipe (the chain goes on to prune / pad ISeq: SeqRecArch.sure_validpipe) emits one row per HELD-OUT ITEM with a growing
                # history -- seq = seen + unseen[:k], unseen = (positive,), seen unchanged (HSTU/sampler.py:101-122); a leave-one-out split has
                # one held-out item per user and nothing changes, a ratio split (*_ROU) gets len(unseen) rows per user
                seq_pipe = any(n in ("lprune", "lpad", "add") and R["ISeq"] in tuple(k.get("fields", ())) for n, k in self.ops)
                if seq_pipe and any(len(t) > 1 for t in row[R["IUnseen"]]):
                    us, hs, ts, ss = [], [], [], []
                    for x, h, t, s in zip(u.tolist(), hist, row[R["IUnseen"]], row[R["ISeen"]]):
                        for k in range(len(t)):
                            us.append(x); hs.append(np.concatenate([h, t[:k]])); ts.append(t[k:k + 1]); ss.append(s)
                    u = np.asarray(us, np.int64)
                    row[R["User"]], row[R["ISeq"]], row[R["IUnseen"]], row[R["ISeen"]] = u, hs, ts, ss
                if kw["ranking"] == "pool":                              # the targets first, then K sampled unseen items
                    neg = self._unseen(u, (len(u), kw["k"]))
                    row[R["IUnseen"]] = [np.concatenate([t, n]) for t, n in zip(row[R["IUnseen"]], neg)]
            elif name == "add":
                for f in kw["fields"]:
                    row[f] = [a + kw["offset"] for a in row[f]] if isinstance(row[f], list) else row[f] + kw["offset"]
            elif name in ("lprune", "rprune"):
                for f in kw["fields"]:
                    row[f] = [a[-kw["maxlen"]:] if name == "lprune" else a[:kw["maxlen"]] for a in row[f]]
            elif name in ("lpad", "rpad"):
                for f in kw["fields"]:
                    rows = row[f]
                    tail = rows[0].shape[1:] if len(rows) else ()
                    out = np.full((len(rows), kw["maxlen"]) + tuple(tail), kw["value"], np.int64)
                    for i, a in enumerate(rows):
                        a = a[-kw["maxlen"]:] if name == "lpad" else a[:kw["maxlen"]]
                        if len(a):
                            if name == "lpad":
                                out[i, kw["maxlen"] - len(a):] = a
                            else:
                                out[i, :len(a)] = a
                    row[f] = out
        row.pop("_seq", None)
        if R:
            row[R["Size"]] = len(seeds)
        return row

    def _finish(self, row):
        if not self.as_tensor:
            return row
        out = {}
        for k, v in row.items():
            if isinstance(v, np.ndarray):
                out[k] = torch.from_numpy(np.ascontiguousarray(v))
            elif isinstance(v, list):
                out[k] = [a.tolist() if isinstance(a, np.ndarray) else a for a in v]     # ragged: lists (ISeen / IUnseen)
            else:
                out[k] = v
        return out

    def __iter__(self):
        if self.device is not None and self.device.type == "cuda":
            from recboard_amd import sampler
            dev = sampler.device_pipe(self, fused=getattr(self, "fused", False))
            if dev is not None:
                yield from dev
                return
        seeds = self._seeds()
        B = self.batch_size or len(seeds)
        for i in range(0, len(seeds), B):
            yield self._finish(self._rows(seeds[i:i + B]))
