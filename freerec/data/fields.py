"""`freerec.data.fields`: a Field names a column (tags + cardinality) and -- being an nn.Module -- owns what a model hangs on it
(`self.Item.add_module("embeddings", nn.Embedding(self.Item.count, D))`, SASRec/main.py:70-77); it is the KEY of a batch dict
(`data[self.ISeq]`).  `fork(tag)` is the same column in another role: equal name, more tags, its own hash."""
import torch
import torch.nn as nn


class Field(nn.Module):
    def __init__(self, name, *tags, count=None):
        super().__init__()
        self.name = str(name)
        self.tags = frozenset(tags)
        self.count = count

    def fork(self, *tags):
        f = Field(self.name, *(self.tags | frozenset(tags)), count=self.count)
        return f

    def match(self, *tags):
        return all(t in self.tags for t in tags)

    def match_any(self, *tags):
        return any(t in self.tags for t in tags)

    def __hash__(self):
        return hash((self.name, self.tags))

    def __eq__(self, other):
        return isinstance(other, Field) and self.name == other.name and self.tags == other.tags

    def __repr__(self):
        return f"Field({self.name}: {'|'.join(sorted(self.tags))}, count={self.count})"

    def to_csr(self, rows):
        """Ragged id lists (one per user) -> a [B, count] sparse CSR indicator, as `Coach.evaluate` builds its seen / target
        matrices (UniSRec/main.py:411-417: `self.Item.to_csr(data[self.ISeen]).to(self.device).to_dense().bool()`)."""
        ptr = torch.zeros(len(rows) + 1, dtype=torch.int64)
        ptr[1:] = torch.cumsum(torch.tensor([len(r) for r in rows], dtype=torch.int64), 0)
        col = torch.cat([torch.as_tensor(sorted(set(map(int, r))), dtype=torch.int64) for r in rows]) if len(rows) else torch.zeros(0, dtype=torch.int64)
        ptr[1:] = torch.cumsum(torch.tensor([len(set(map(int, r))) for r in rows], dtype=torch.int64), 0)
        return torch.sparse_csr_tensor(ptr, col, torch.ones(col.numel()), size=(len(rows), int(self.count)))


class FieldModuleList(nn.ModuleList):
    def match(self, *tags):
        return FieldModuleList([f for f in self if f.match(*tags)])

    def match_not(self, *tags):
        return FieldModuleList([f for f in self if not f.match_any(*tags)])

    def match_all(self, *tags):
        return self.match(*tags)

    def match_any(self, *tags):
        return FieldModuleList([f for f in self if f.match_any(*tags)])

    def __getitem__(self, idx):
        if isinstance(idx, (tuple, str)):
            want = idx if isinstance(idx, tuple) else (idx,)
            for f in self:
                if f.match(*want):
                    return f
            raise KeyError(f"no field with tags {want}")
        return super().__getitem__(idx)
