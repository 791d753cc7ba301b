"""`freerec.data.datasets`: `RecDataSet(root, filedir, tasktag=)` (SASRec/main.py:264-269) with `.fields`, `.train() / .valid() / .test()`
splits, graph views of the training interactions and the source factories of the datapipe chains.

On disk (no dataset ships with the reference; the layout is this package's own): `<root>/<filedir>/{train,valid,test}.txt` (also looked
for under `<root>/Processed/<filedir>`), tab-separated with a header row naming the columns -- USER, ITEM and optionally TIMESTAMP /
RATING for the interaction datasets (ids already 0-based and dense), any columns plus LABEL for prediction datasets.  In memory:
`RecDataSet.from_sequences(seqs, num_items)` (leave-one-out: last item = test, the one before = valid) and
`RecDataSet.from_splits(train, valid, test, num_users, num_items)`; `PredictionRecDataSet.from_columns(...)`."""
import os
import types

import numpy as np
import torch

from . import tags as T
from .fields import Field, FieldModuleList
from .postprocessing import Pipe


def _read_tsv(path):
    with open(path) as f:
        header = f.readline().rstrip("\n").split("\t")
    raw = np.loadtxt(path, delimiter="\t", skiprows=1, ndmin=2)
    return {h.strip().upper(): raw[:, i] for i, h in enumerate(header)}


class _Split:
    """One split of a dataset (`dataset.train()`): graph views and datapipe sources."""

    def __init__(self, ds, mode):
        self.ds, self.mode = ds, mode

    # ---- sizes
    @property
    def datasize(self):
        return len(self.ds.splits[self.mode]["USER"]) if "USER" in self.ds.splits[self.mode] else len(next(iter(self.ds.splits[self.mode].values())))

    # ---- graph views of THIS split's interactions (the scripts only ask the training split)
    def _ui(self):
        s = self.ds.splits[self.mode]
        return np.asarray(s["USER"], np.int64), np.asarray(s["ITEM"], np.int64)

    def to_bigraph(self, edge_type="u2i"):
        u, i = self._ui()
        return {edge_type: types.SimpleNamespace(edge_index=torch.from_numpy(np.stack([u, i])))}

    def to_graph(self):
        """Users and items as one node set (items offset by the user count), both directions (NGCF/main.py:76)."""
        u, i = self._ui()
        i = i + self.ds.num_users
        return types.SimpleNamespace(edge_index=torch.from_numpy(np.stack([np.concatenate([u, i]), np.concatenate([i, u])])))

    def to_normalized_adj(self, normalization="sym"):
        """The bipartite interaction graph as a normalised [U + N, U + N] sparse CSR matrix, no self loops (LightGCN/main.py:47-49;
        NGCF/main.py:77 adds them itself, so the default has none)."""
        from .. import graph
        ei = self.to_graph().edge_index
        n = self.ds.num_users + self.ds.num_items
        key = torch.unique(ei[0] * n + ei[1])            # distinct edges, sorted by (row, col)
        ei = torch.stack([key // n, key % n])
        ei, w = graph.to_normalized(ei, None, normalization, n)
        crow = torch.zeros(n + 1, dtype=torch.int64)
        crow[1:] = torch.cumsum(torch.bincount(ei[0], minlength=n), 0)
        return torch.sparse_csr_tensor(crow, ei[1].contiguous(), w.contiguous(), size=(n, n))

    # ---- sources
    def shuffled_seqs_source(self, maxlen=None, keep_at_least_itself=True):
        return Pipe(self, "shuffled_seqs", dict(maxlen=maxlen))

    def ordered_seqs_source(self, maxlen=None):
        return Pipe(self, "ordered_seqs", dict(maxlen=maxlen))

    def choiced_user_ids_source(self):
        return Pipe(self, "choiced_user_ids", {})

    def ordered_user_ids_source(self):
        return Pipe(self, "ordered_user_ids", {})

    def shuffled_pairs_source(self):
        return Pipe(self, "shuffled_pairs", {})

    def shuffled_inter_source(self):
        return Pipe(self, "shuffled_inter", {})

    def ordered_inter_source(self):
        return Pipe(self, "ordered_inter", {})


class RecDataSet:
    TASK = T.MATCHING

    def __init__(self, root=None, filedir=None, tasktag=None, *, splits=None, fields=None, cfg=None, name=None):
        self.root, self.filedir, self.tasktag = root, filedir, tasktag or self.TASK
        self.name = name or filedir or type(self).__name__
        if splits is None:
            splits = self._load(root, filedir)
        self.splits = {k: {c: np.asarray(v) for c, v in s.items()} for k, s in splits.items()}
        self.fields = FieldModuleList(fields if fields is not None else self._infer_fields())
        self._build_indices()

    # ---- construction
    @classmethod
    def from_splits(cls, train, valid, test, num_users=None, num_items=None, name="in-memory"):
        """train / valid / test: (users, items) arrays (or dicts of columns)."""
        mk = lambda s: s if isinstance(s, dict) else {"USER": np.asarray(s[0], np.int64), "ITEM": np.asarray(s[1], np.int64)}  # noqa: E731
        ds = cls.__new__(cls)
        sp = {"train": mk(train), "valid": mk(valid), "test": mk(test)}
        U = num_users if num_users is not None else int(max(int(s["USER"].max()) for s in sp.values() if len(s["USER"]))) + 1
        N = num_items if num_items is not None else int(max(int(s["ITEM"].max()) for s in sp.values() if len(s["ITEM"]))) + 1
        fields = [Field("USER", T.USER, T.ID, count=U), Field("ITEM", T.ITEM, T.ID, count=N)]
        RecDataSet.__init__(ds, None, None, None, splits=sp, fields=fields, name=name)
        return ds

    @classmethod
    def from_sequences(cls, seqs, num_items, name="in-memory"):
        """Per-user chronological item sequences -> leave-one-out splits (users with fewer than 3 items keep everything in train)."""
        tr_u, tr_i, va_u, va_i, te_u, te_i = [], [], [], [], [], []
        for u, s in enumerate(seqs):
            s = np.asarray(s, np.int64)
            cut = len(s) - 2 if len(s) >= 3 else len(s)
            tr_u.append(np.full(cut, u)); tr_i.append(s[:cut])
            if len(s) >= 3:
                va_u.append(u); va_i.append(s[-2]); te_u.append(u); te_i.append(s[-1])
        cat = lambda a: np.concatenate(a) if a else np.zeros(0, np.int64)  # noqa: E731
        return cls.from_splits((cat(tr_u), cat(tr_i)), (np.asarray(va_u, np.int64), np.asarray(va_i, np.int64)),
                               (np.asarray(te_u, np.int64), np.asarray(te_i, np.int64)), len(seqs), num_items, name)

    def _load(self, root, filedir):
        cands = [os.path.join(root or ".", filedir or ""), os.path.join(root or ".", "Processed", filedir or "")]
        for d in cands:
            if all(os.path.exists(os.path.join(d, f"{m}.txt")) for m in ("train", "valid", "test")):
                return {m: _read_tsv(os.path.join(d, f"{m}.txt")) for m in ("train", "valid", "test")}
        raise FileNotFoundError(f"freerec (recengine surface): no dataset at {cands[0]!r} (train.txt / valid.txt / test.txt, tab-separated with a "
                                "header row: USER, ITEM[, TIMESTAMP]); see freerec/data/datasets.py")

    def _infer_fields(self):
        cols = list(self.splits["train"].keys())
        out = []
        for c in cols:
            vals = np.concatenate([np.asarray(self.splits[m][c]) for m in self.splits if c in self.splits[m]])
            if c == "USER":
                out.append(Field(c, T.USER, T.ID, count=int(vals.max()) + 1 if vals.size else 0))
            elif c == "ITEM":
                out.append(Field(c, T.ITEM, T.ID, count=int(vals.max()) + 1 if vals.size else 0))
            elif c == "TIMESTAMP":
                out.append(Field(c, T.TIMESTAMP))
            elif c == "LABEL":
                out.append(Field(c, T.LABEL))
            elif np.allclose(vals, np.round(vals)):
                out.append(Field(c, T.FEATURE, T.SPARSE, T.EMBED, count=int(vals.max()) + 1 if vals.size else 0))
            else:
                out.append(Field(c, T.FEATURE, T.DENSE, T.EMBED))
        return out

    def _build_indices(self):
        if "USER" not in self.splits["train"] or "ITEM" not in self.splits["train"]:
            return
        self.num_users = int(self.fields[T.USER, T.ID].count)
        self.num_items = int(self.fields[T.ITEM, T.ID].count)
        U = self.num_users
        self._seqs = {}
        for mode in ("train", "valid", "test"):
            s = self.splits[mode]
            u, i = np.asarray(s["USER"], np.int64), np.asarray(s["ITEM"], np.int64)
            order = np.lexsort((np.asarray(s["TIMESTAMP"]), u)) if "TIMESTAMP" in s else np.argsort(u, kind="stable")
            u, i = u[order], i[order]
            ptr = np.zeros(U + 1, np.int64)
            np.cumsum(np.bincount(u, minlength=U), out=ptr[1:])
            self._seqs[mode] = (ptr, i)
        # sorted (user, item) keys of the training interactions: the negative samplers' "seen" test
        ptr, items = self._seqs["train"]
        users = np.repeat(np.arange(U), np.diff(ptr))
        self._seen_keys = np.unique(users * self.num_items + items)

    # ---- access
    def train(self):
        return _Split(self, "train")

    def valid(self):
        return _Split(self, "valid")

    def test(self):
        return _Split(self, "test")

    def seq(self, mode, u):
        ptr, items = self._seqs[mode]
        return items[ptr[u]:ptr[u + 1]]

    @property
    def trainsize(self):
        return len(self.splits["train"]["USER"]) if "USER" in self.splits["train"] else len(next(iter(self.splits["train"].values())))

    def summary(self):
        return {"dataset": self.name, "users": getattr(self, "num_users", None), "items": getattr(self, "num_items", None),
                **{m: _Split(self, m).datasize for m in self.splits}}

    def to(self, device):
        return self


class MatchingRecDataSet(RecDataSet):
    TASK = T.MATCHING


class NextItemRecDataSet(RecDataSet):
    TASK = T.NEXTITEM


class PredictionRecDataSet(RecDataSet):
    """Row-wise prediction data (DeepFM/main.py:282-290): feature columns + LABEL."""
    TASK = T.PREDICTION

    @classmethod
    def from_columns(cls, train, valid, test, fields, name="in-memory"):
        ds = cls.__new__(cls)
        RecDataSet.__init__(ds, None, None, None, splits={"train": train, "valid": valid, "test": test}, fields=fields, name=name)
        return ds
