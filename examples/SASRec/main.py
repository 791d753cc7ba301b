"""SASRec as a client of the `freerec` surface -- the recengine's own model file, laid out the way RecBoard's model scripts are
(`cfg = freerec.parser.Parser()` at module level, a `SeqRecArch` subclass with `sure_trainpipe / encode / fit / recommend_from_*`, a
`Coach` subclass with the step loop, `main()`), with the reference's parameter names (`Item.embeddings`, `Position`, `attnLNs`,
`attnLayers`, `fwdLNs`, `fwdLayers.{conv1,conv2}`, `lastLN`), so checkpoints interchange with RecBoard's SASRec/main.py.

    python examples/SASRec/main.py --root data --dataset MyDataset --epochs 200            # fused engine step (cfg.engine = auto)
    python examples/SASRec/main.py ... --engine module                                     # the torch code below, op by op

Arithmetic (SASRec/main.py:163-193 of the reference): x = E[seq] sqrt(D) + P[0..S), dropout, pads zeroed; per block
x = MHA(LN(x), x, x, causal) + x; y = LN(x); x = FFN(y) + y; pads zeroed; u = LN(x).  BCE / BPR / CE over the non-pad positions."""
import freerec
import torch
import torch.nn as nn

freerec.declare(version="1.0.1")

cfg = freerec.parser.Parser()
cfg.add_argument("--maxlen", type=int, default=50)
cfg.add_argument("--num-heads", type=int, default=1)
cfg.add_argument("--num-blocks", type=int, default=2)
cfg.add_argument("--embedding-dim", type=int, default=64)
cfg.add_argument("--dropout-rate", type=float, default=0.2)
cfg.add_argument("--loss", type=str, choices=("BPR", "BCE", "CE"), default="BCE")
cfg.set_defaults(description="SASRec", root="../../data", dataset="Amazon2014Beauty_550_LOU", epochs=200, batch_size=256,
                 optimizer="adam", lr=1e-3, weight_decay=0.0, seed=1)
cfg.compile()


class FeedForward(nn.Module):
    """Two position-wise maps as Conv1d(k = 1), dropout after each, residual onto the input."""

    def __init__(self, dim, p):
        super().__init__()
        self.conv1, self.dropout1 = nn.Conv1d(dim, dim, kernel_size=1), nn.Dropout(p)
        self.relu = nn.ReLU()
        self.conv2, self.dropout2 = nn.Conv1d(dim, dim, kernel_size=1), nn.Dropout(p)

    def forward(self, y):
        h = self.relu(self.dropout1(self.conv1(y.transpose(-1, -2))))
        return self.dropout2(self.conv2(h)).transpose(-1, -2) + y


class SASRec(freerec.models.SeqRecArch):
    def __init__(self, dataset):
        super().__init__(dataset)
        D, S = cfg.embedding_dim, cfg.maxlen
        self.num_blocks = cfg.num_blocks
        self.Item.add_module("embeddings", nn.Embedding(self.Item.count + self.NUM_PADS, D, padding_idx=self.PADDING_VALUE))
        self.Position = nn.Embedding(S, D)
        self.embdDropout = nn.Dropout(cfg.dropout_rate)
        self.register_buffer("positions", torch.arange(S, dtype=torch.long).unsqueeze(0))
        self.register_buffer("attnMask", torch.ones(S, S, dtype=torch.bool).triu(1))
        self.attnLNs = nn.ModuleList(nn.LayerNorm(D, eps=1e-8) for _ in range(self.num_blocks))
        self.attnLayers = nn.ModuleList(nn.MultiheadAttention(D, cfg.num_heads, dropout=cfg.dropout_rate, batch_first=True)
                                        for _ in range(self.num_blocks))
        self.fwdLNs = nn.ModuleList(nn.LayerNorm(D, eps=1e-8) for _ in range(self.num_blocks))
        self.fwdLayers = nn.ModuleList(FeedForward(D, cfg.dropout_rate) for _ in range(self.num_blocks))
        self.lastLN = nn.LayerNorm(D, eps=1e-8)
        make = {"BCE": freerec.criterions.BCELoss4Logits, "BPR": freerec.criterions.BPRLoss, "CE": freerec.criterions.CrossEntropy4Logits}
        self.criterion = make[cfg.loss](reduction="mean")
        self.reset_parameters()

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.xavier_normal_(m.weight)
                if getattr(m, "bias", None) is not None:
                    nn.init.zeros_(m.bias)

    def sure_trainpipe(self, maxlen, batch_size):
        return (self.dataset.train().shuffled_seqs_source(maxlen=maxlen)
                .seq_train_yielding_pos_(start_idx_for_target=1, end_idx_for_input=-1)
                .seq_train_sampling_neg_(num_negatives=1)
                .add_(offset=self.NUM_PADS, modified_fields=(self.ISeq,))
                .lpad_(maxlen, modified_fields=(self.ISeq, self.IPos, self.INeg), padding_value=self.PADDING_VALUE)
                .batch_(batch_size).tensor_())

    def encode(self, data):
        seq = data[self.ISeq]
        pad = (seq == self.PADDING_VALUE).unsqueeze(-1)
        x = self.Item.embeddings(seq) * (cfg.embedding_dim ** 0.5) + self.Position(self.positions)
        x = self.embdDropout(x).masked_fill(pad, 0.0)
        for l in range(self.num_blocks):
            q = self.attnLNs[l](x)
            x = self.attnLayers[l](q, x, x, attn_mask=self.attnMask, need_weights=False)[0] + x
            x = self.fwdLayers[l](self.fwdLNs[l](x)).masked_fill(pad, 0.0)
        return self.lastLN(x), self.Item.embeddings.weight[self.NUM_PADS:]

    def fit(self, data):
        users, items = self.encode(data)
        keep = data[self.ISeq] != self.PADDING_VALUE
        users = users[keep]
        if cfg.loss in ("BCE", "BPR"):
            pos = (users * items[data[self.IPos][keep]]).sum(-1)
            neg = (users * items[data[self.INeg][keep]]).sum(-1)
            if cfg.loss == "BPR":
                return {"rec_loss": self.criterion(pos, neg)}
            return {"rec_loss": self.criterion(pos, torch.ones_like(pos)) + self.criterion(neg, torch.zeros_like(neg))}
        return {"rec_loss": self.criterion(users @ items.t(), data[self.IPos][keep])}

    def recommend_from_full(self, data):
        users, items = self.encode(data)
        return users[:, -1, :] @ items.t()

    def recommend_from_pool(self, data):
        users, items = self.encode(data)
        return torch.einsum("BD,BKD->BK", users[:, -1, :], items[data[self.IUnseen]])


class CoachForSASRec(freerec.launcher.Coach):
    def train_per_epoch(self, epoch):
        for data in self.dataloader:
            data = self.dict_to_device(data)
            loss = self.model(data)["rec_loss"]
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()
            self.monitor(loss.item(), n=len(data[self.User]), reduction="mean", mode="train", pool=["LOSS"])


def main():
    try:
        dataset = getattr(freerec.data.datasets, cfg.dataset)(root=cfg.root)
    except AttributeError:
        dataset = freerec.data.datasets.RecDataSet(cfg.root, cfg.dataset, tasktag=cfg.tasktag)
    model = SASRec(dataset)
    coach = CoachForSASRec(dataset=dataset, trainpipe=model.sure_trainpipe(cfg.maxlen, cfg.batch_size),
                           validpipe=model.sure_validpipe(cfg.maxlen, ranking=cfg.ranking),
                           testpipe=model.sure_testpipe(cfg.maxlen, ranking=cfg.ranking), model=model, cfg=cfg)
    return coach.fit()


if __name__ == "__main__":
    main()
