"""`freerec.models`: the architecture base classes the scripts subclass (`class SASRec(freerec.models.SeqRecArch)`, SASRec/main.py:53).

A RecSysArch holds the dataset and its fields, names the roles a batch dict is keyed by (User, Item, ISeq, IPos, INeg, IUnseen, ISeen,
Label, Size), and dispatches `model(data, ranking=)`: `fit(data)` while training, `recommend_from_full / _pool(data)` in evaluation
(evidenced by DIGER/main.py:413-415).  `sure_validpipe / sure_testpipe` are the inherited evaluation pipes, with the signatures the
scripts call them by: Seq `(maxlen, ranking=)` (SASRec/main.py:275-276), Gen `(ranking)` (MF-BPR/main.py:145-146), Pred `()`."""
import torch
import torch.nn as nn

from ..data import tags as T
from ..data.fields import Field, FieldModuleList
from . import nn as nn_  # noqa: F401  (freerec.models.nn.Unsqueeze)


class RecSysArch(nn.Module):
    NUM_PADS = 0
    PADDING_VALUE = 0

    def __init__(self, dataset):
        super().__init__()
        self.dataset = dataset
        # (not a registered submodule: the fields a model uses are registered under their role names -- `Item.embeddings.weight`)
        object.__setattr__(self, "fields", dataset.fields if isinstance(dataset.fields, FieldModuleList) else FieldModuleList(dataset.fields))
        try:
            self.User = self.fields[T.USER, T.ID]
            self.Item = self.fields[T.ITEM, T.ID]
            self.ISeq = self.Item.fork(T.SEQUENCE)
            self.IPos = self.Item.fork(T.POSITIVE)
            self.INeg = self.Item.fork(T.NEGATIVE)
            self.IUnseen = self.Item.fork(T.UNSEEN)
            self.ISeen = self.Item.fork(T.SEEN)
        except KeyError:
            pass
        try:
            self.Label = self.fields[T.LABEL]
        except KeyError:
            pass
        self.Size = Field("SIZE", T.SIZE)

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    def reset_ranking_buffers(self):
        """Called once before every evaluation pass (MF-BPR/main.py:95-99 clones its tables there)."""

    def fit(self, data):
        raise NotImplementedError

    def recommend_from_full(self, data):
        raise NotImplementedError

    def recommend_from_pool(self, data):
        raise NotImplementedError

    def forward(self, data, ranking: str = "full"):
        if self.training:
            return self.fit(data)
        if ranking == "full":
            return self.recommend_from_full(data)
        if ranking == "pool":
            return self.recommend_from_pool(data)
        raise NotImplementedError(f"`ranking` should be 'full' or 'pool' but {ranking} received ...")

    def sure_trainpipe(self, *a, **k):
        raise NotImplementedError


class GenRecArch(RecSysArch):
    def sure_validpipe(self, ranking: str = "full", batch_size: int = 512):
        return self.dataset.valid().ordered_user_ids_source().valid_sampling_(ranking).batch_(batch_size).tensor_()

    def sure_testpipe(self, ranking: str = "full", batch_size: int = 512):
        return self.dataset.test().ordered_user_ids_source().test_sampling_(ranking).batch_(batch_size).tensor_()


class SeqRecArch(RecSysArch):
    NUM_PADS = 1
    PADDING_VALUE = 0

    def sure_validpipe(self, maxlen: int, ranking: str = "full", batch_size: int = 512):
        return (self.dataset.valid().ordered_user_ids_source().valid_sampling_(ranking)
                .lprune_(maxlen, modified_fields=(self.ISeq,)).add_(offset=self.NUM_PADS, modified_fields=(self.ISeq,))
                .lpad_(maxlen, modified_fields=(self.ISeq,), padding_value=self.PADDING_VALUE).batch_(batch_size).tensor_())

    def sure_testpipe(self, maxlen: int, ranking: str = "full", batch_size: int = 512):
        return (self.dataset.test().ordered_user_ids_source().test_sampling_(ranking)
                .lprune_(maxlen, modified_fields=(self.ISeq,)).add_(offset=self.NUM_PADS, modified_fields=(self.ISeq,))
                .lpad_(maxlen, modified_fields=(self.ISeq,), padding_value=self.PADDING_VALUE).batch_(batch_size).tensor_())


class PredRecArch(RecSysArch):
    def sure_validpipe(self, batch_size: int = 4096):
        return self.dataset.valid().ordered_inter_source().batch_(batch_size).tensor_()

    def sure_testpipe(self, batch_size: int = 4096):
        return self.dataset.test().ordered_inter_source().batch_(batch_size).tensor_()
