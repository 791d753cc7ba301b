"""Minimal stand-in for the third-party `freerec` package (pinned 1.0.1 by the
reference, NOT vendored under /root/reference, NOT installable offline).

TEST SCAFFOLDING ONLY.  Used by ``make_golden.py`` in the development container
to import the reference model classes (``/root/reference/<Model>/main.py``) and
dump golden vectors.  It is written from the *call sites* in the reference
 and is not a copy of freerec.  Nothing in the product
(`recboard_amd/`) imports it.

Parity note: the criterion formulas below are restated from call-site evidence
("parity unpinned" at the freerec boundary):
  BPRLoss            = softplus(neg - pos)             (MF-BPR/main.py:88-91; untrained value = ln 2)
  BCELoss4Logits     = BCE-with-logits                 (SASRec/main.py:211-214, DeepFM/main.py:214)
  CrossEntropy4Logits= F.cross_entropy                 (SASRec/main.py:217-219)
  regularize(l2)     = sum ||p||^2 / 2                 (LightGCN/main.py:99-106, mirrors MF-BPR/main.py:70-76)
"""
import argparse
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


def build(argv_defaults=None):
    fr = types.ModuleType("freerec")

    def declare(version=None):
        return None

    fr.declare = declare

    # ---------------- parser ----------------
    class Parser:
        def __init__(self):
            self._p = argparse.ArgumentParser()
            self._defaults = dict(
                ranking="full", tasktag=None, device="cpu", monitors=[], eval_freq=5,
                optim_first_moment_decay=0.9, optim_second_moment_decay=0.999,
                adam_beta1=0.9, adam_beta2=0.999, sgd_momentum=0.9, sgd_nesterov=False,
                lr_scheduler={}, config=None,
            )

        def add_argument(self, *a, **k):
            self._p.add_argument(*a, **k)

        def set_defaults(self, **k):
            self._defaults.update(k)

        def compile(self):
            ns, _ = self._p.parse_known_args(sys.argv[1:])
            for k, v in self._defaults.items():
                setattr(self, k, v)
            for k, v in vars(ns).items():
                setattr(self, k, v)
            for k, v in (argv_defaults or {}).items():
                setattr(self, k, v)

        def get(self, k, d=None):
            return getattr(self, k, d)

    fr.parser = types.ModuleType("freerec.parser")
    fr.parser.Parser = Parser

    # ---------------- data ----------------
    data = types.ModuleType("freerec.data")
    tags = types.ModuleType("freerec.data.tags")
    for t in ("USER", "ITEM", "ID", "SEQUENCE", "POSITIVE", "NEGATIVE", "UNSEEN", "SEEN",
              "EMBED", "LABEL", "TIMESTAMP", "SIZE"):
        setattr(tags, t, t)
    data.tags = tags

    class Field(nn.Module):
        def __init__(self, name, *ftags, count=None):
            super().__init__()
            self.name = name
            self.tags = set(ftags)
            self.count = count

        def fork(self, *ftags):
            f = Field(self.name, *(self.tags | set(ftags)), count=self.count)
            return f

        def match(self, *ftags):
            return all(t in self.tags for t in ftags)

        def __hash__(self):
            return hash((self.name, tuple(sorted(self.tags))))

        def __eq__(self, other):
            return isinstance(other, Field) and hash(self) == hash(other)

    class FieldModuleList(nn.ModuleList):
        def match(self, *ftags):
            return FieldModuleList([f for f in self if f.match(*ftags)])

        def match_not(self, *ftags):
            return FieldModuleList([f for f in self if not any(t in f.tags for t in ftags)])

        def __getitem__(self, idx):
            if isinstance(idx, tuple) or isinstance(idx, str):
                idx = idx if isinstance(idx, tuple) else (idx,)
                for f in self:
                    if f.match(*idx):
                        return f
                raise KeyError(idx)
            return super().__getitem__(idx)

    fields_mod = types.ModuleType("freerec.data.fields")
    fields_mod.Field = Field
    fields_mod.FieldModuleList = FieldModuleList
    data.fields = fields_mod

    class _Split:
        def __init__(self, ds):
            self.ds = ds

        def to_normalized_adj(self, normalization="sym"):
            return self.ds.adj

        def to_graph(self):
            """The training interactions as an undirected bipartite graph: edge_index [2, 2E] (NGCF/main.py:76)."""
            return types.SimpleNamespace(edge_index=self.ds.edge_index)

        def to_bigraph(self, edge_type="u2i"):
            """The training interactions as user -> item edges: {"u2i": edge_index [2, E]} with item ids 0-based (SGL/main.py:55-58)."""
            return {edge_type: types.SimpleNamespace(edge_index=self.ds.u2i)}

    class RecDataSet:
        """Toy dataset: carries fields (+ a prebuilt normalised adjacency for LightGCN)."""

        def __init__(self, fields, adj=None, edge_index=None, u2i=None):
            self.fields = FieldModuleList(fields)
            self.adj = adj
            self.edge_index = edge_index
            self.u2i = u2i

        def train(self):
            return _Split(self)

    ds_mod = types.ModuleType("freerec.data.datasets")
    ds_mod.RecDataSet = RecDataSet
    ds_mod.NextItemRecDataSet = RecDataSet
    ds_mod.PredictionRecDataSet = RecDataSet
    data.datasets = ds_mod
    fr.data = data

    # ---------------- graph (NGCF/main.py:76-87: self loops + "left" normalisation; stated here, parity unpinned like the rest of freerec) ----
    def add_self_loops(edge_index, num_nodes=None):
        n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
        loops = torch.arange(n, dtype=edge_index.dtype).repeat(2, 1)
        return torch.cat((edge_index, loops), dim=1), None

    def to_normalized(edge_index, edge_weight=None, normalization="sym"):
        n = int(edge_index.max()) + 1
        w = torch.ones(edge_index.shape[1]) if edge_weight is None else edge_weight
        deg = torch.zeros(n).index_add_(0, edge_index[0], w)
        if normalization == "left":        # D^-1 A
            w = w / deg[edge_index[0]]
        elif normalization == "right":
            w = w / deg[edge_index[1]]
        else:                              # (a node whose edges all carry weight 0 -- SGL's dropped edges -- has degree 0: its entries stay 0)
            dinv = torch.where(deg > 0, deg.rsqrt(), torch.zeros_like(deg))
            w = w * dinv[edge_index[0]] * dinv[edge_index[1]]
        return edge_index, w

    def to_undirected(edge_index, edge_weight=None, num_nodes=None):
        """Both directions of every edge, weights carried along (SGL/main.py:103-105)."""
        ei = torch.cat((edge_index, edge_index.flip(0)), dim=1)
        return ei, (None if edge_weight is None else torch.cat((edge_weight, edge_weight)))

    def to_adjacency(edge_index, edge_weight, num_nodes):
        return torch.sparse_coo_tensor(edge_index, edge_weight, (num_nodes, num_nodes)).coalesce().to_sparse_csr()

    graph_mod = types.ModuleType("freerec.graph")
    graph_mod.add_self_loops, graph_mod.to_normalized, graph_mod.to_adjacency = add_self_loops, to_normalized, to_adjacency
    graph_mod.to_undirected = to_undirected
    fr.graph = graph_mod

    # ---------------- models ----------------
    class RecSysArch(nn.Module):
        NUM_PADS = 0
        PADDING_VALUE = 0

        def __init__(self, dataset):
            super().__init__()
            self.dataset = dataset
            object.__setattr__(self, "fields", dataset.fields)  # not a registered submodule (avoids alias keys)
            try:
                self.User = self.fields[tags.USER, tags.ID]
                self.Item = self.fields[tags.ITEM, tags.ID]
                self.ISeq = self.Item.fork(tags.SEQUENCE)
                self.IPos = self.Item.fork(tags.POSITIVE)
                self.INeg = self.Item.fork(tags.NEGATIVE)
                self.IUnseen = self.Item.fork(tags.UNSEEN)
                self.ISeen = self.Item.fork(tags.SEEN)
            except KeyError:
                pass
            try:
                self.Label = self.fields[tags.LABEL]
            except KeyError:
                pass
            self.Size = Field("SIZE", tags.SIZE)

        @property
        def device(self):
            return next(self.parameters()).device

        def forward(self, data, ranking="train"):
            if self.training:
                return self.fit(data)
            if ranking == "full":
                return self.recommend_from_full(data)
            return self.recommend_from_pool(data)

    class GenRecArch(RecSysArch):
        pass

    class SeqRecArch(RecSysArch):
        NUM_PADS = 1
        PADDING_VALUE = 0

    class PredRecArch(RecSysArch):
        pass

    class Unsqueeze(nn.Module):
        def __init__(self, dim):
            super().__init__()
            self.dim = dim

        def forward(self, x):
            return x.unsqueeze(self.dim)

    models = types.ModuleType("freerec.models")
    models.RecSysArch, models.GenRecArch, models.SeqRecArch, models.PredRecArch = (
        RecSysArch, GenRecArch, SeqRecArch, PredRecArch)
    models.nn = types.ModuleType("freerec.models.nn")
    models.nn.Unsqueeze = Unsqueeze
    fr.models = models

    # ---------------- criterions ----------------
    class BaseCriterion(nn.Module):
        def __init__(self, reduction="mean"):
            super().__init__()
            self.reduction = reduction

        def _reduce(self, x):
            if self.reduction == "mean":
                return x.mean()
            if self.reduction == "sum":
                return x.sum()
            return x

        @staticmethod
        def regularize(params, rtype="l2"):
            params = [params] if isinstance(params, torch.Tensor) else params
            if rtype == "l1":
                return sum(p.abs().sum() for p in params)
            return sum(p.pow(2).sum() for p in params) / 2

    class BPRLoss(BaseCriterion):
        def forward(self, pos, neg):
            return self._reduce(F.softplus(neg - pos))

    class BCELoss4Logits(BaseCriterion):
        def forward(self, logits, targets):
            return F.binary_cross_entropy_with_logits(
                logits, targets.to(logits.dtype), reduction=self.reduction)

    class CrossEntropy4Logits(BaseCriterion):
        def forward(self, logits, targets):
            return F.cross_entropy(logits, targets, reduction=self.reduction)

    def cross_entropy_with_logits(logits, targets, reduction="mean"):
        return F.cross_entropy(logits, targets, reduction=reduction)

    crit = types.ModuleType("freerec.criterions")
    crit.BaseCriterion, crit.BPRLoss, crit.BCELoss4Logits, crit.CrossEntropy4Logits = (
        BaseCriterion, BPRLoss, BCELoss4Logits, CrossEntropy4Logits)
    crit.cross_entropy_with_logits = cross_entropy_with_logits
    fr.criterions = crit

    # ---------------- launcher ----------------
    class Coach:
        def __init__(self, *a, **k):
            pass

    launcher = types.ModuleType("freerec.launcher")
    launcher.Coach = Coach
    fr.launcher = launcher
    fr.utils = types.ModuleType("freerec.utils")
    fr.utils.debugLogger = lambda *a, **k: None
    fr.utils.infoLogger = lambda *a, **k: None

    mods = {
        "freerec": fr, "freerec.parser": fr.parser, "freerec.data": data,
        "freerec.data.tags": tags, "freerec.data.fields": fields_mod,
        "freerec.data.datasets": ds_mod, "freerec.models": models,
        "freerec.models.nn": models.nn, "freerec.criterions": crit,
        "freerec.launcher": launcher, "freerec.utils": fr.utils, "freerec.graph": graph_mod,
    }
    return fr, mods
