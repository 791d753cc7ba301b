"""DeepFM as a client of the `freerec` surface -- the recengine's own model file with the reference's structure and parameter names
(per-field `embeddings` / `embeddings_lr`, `fm.lr_layer.bias`, `dnn.{i}.linear / bn`, `dnn.{n}`), so checkpoints interchange with
RecBoard's DeepFM/main.py.

    python examples/DeepFM/main.py --root data --dataset MyDataset              # the DeepFM engine after the adoption probe
    python examples/DeepFM/main.py ... --engine module                          # the torch code below

Arithmetic (DeepFM/main.py:34-124,201-219 of the reference): E = [emb_f(x_f)] [B, F, D]; lr = sum_f w_f[x_f] + b; fm = 1/2 sum_d ((sum_f E)^2 -
sum_f E^2); dnn = MLP(flatten E) with Linear -> [BatchNorm1d] -> ReLU -> Dropout blocks and a final Linear(., 1); logit = lr + fm + dnn; BCE.
Optimizer: two groups -- names containing "embeddings" decay with cfg.embedding_decay, the rest with cfg.weight_decay; gradient norm clipped
at 10; ReduceLROnPlateau on the best monitored value, stepped in front of every epoch."""
import freerec
import torch
import torch.nn as nn
from freerec.data.tags import EMBED, LABEL

freerec.declare(version="1.0.1")

cfg = freerec.parser.Parser()
cfg.add_argument("--embedding-dim", type=int, default=10)
cfg.add_argument("--hidden-dims", type=str, default="400,400,400")
cfg.add_argument("--hidden-dropout-rate", type=float, default=0.1)
cfg.add_argument("--batch-norm", type=eval, default=False)
cfg.add_argument("--embedding-decay", type=float, default=0.05)
cfg.set_defaults(description="DeepFM", root="../../data", dataset="Frappe_x1_BARS", epochs=100, batch_size=2048, optimizer="adam",
                 lr=1e-3, weight_decay=0.0, eval_freq=100, ranking="pool", seed=1)
cfg.compile()


class FirstOrder(nn.Module):
    """sum_f w_f[x_f] + b: one [count_f, 1] table per field, hung on the field as `embeddings_lr`."""

    def __init__(self, fields):
        super().__init__()
        self.input_fields = fields
        for f in fields:
            f.add_module("embeddings_lr", nn.Embedding(f.count, 1))
        self.bias = nn.Parameter(torch.zeros(1))

    def forward(self, data):
        return torch.stack([f.embeddings_lr(data[f]).reshape(-1) for f in self.input_fields], 1).sum(1, keepdim=True) + self.bias


class FM(nn.Module):
    def __init__(self, fields):
        super().__init__()
        self.lr_layer = FirstOrder(fields)

    def forward(self, data, E):
        s = E.sum(1)
        return self.lr_layer(data) + 0.5 * (s * s - (E * E).sum(1)).sum(-1, keepdim=True)


class Block(nn.Module):
    def __init__(self, n_in, n_out, batch_norm, p):
        super().__init__()
        self.linear = nn.Linear(n_in, n_out)
        self.bn = nn.BatchNorm1d(n_out) if batch_norm else nn.Identity()
        self.act = nn.ReLU()
        self.dropout = nn.Dropout(p)

    def forward(self, x):
        return self.dropout(self.act(self.bn(self.linear(x))))


class DeepFM(freerec.models.PredRecArch):
    def __init__(self, dataset):
        super().__init__(dataset)
        self.input_fields = self.fields.match_not(LABEL)
        if len(self.input_fields.match(EMBED)) != len(self.input_fields):
            raise NotImplementedError("this example embeds categorical fields only")
        D = cfg.embedding_dim
        for f in self.input_fields:
            f.add_module("embeddings", nn.Embedding(f.count, D))
        dims = [len(self.input_fields) * D] + [int(h) for h in str(cfg.hidden_dims).split(",")]
        self.dnn = nn.Sequential(*[Block(a, b, cfg.batch_norm, cfg.hidden_dropout_rate) for a, b in zip(dims[:-1], dims[1:])], nn.Linear(dims[-1], 1))
        self.fm = FM(self.input_fields)
        self.criterion = freerec.criterions.BCELoss4Logits(reduction="mean")
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Embedding):
                    m.weight.normal_(0.0, 1e-4)
                elif isinstance(m, nn.Linear):
                    nn.init.xavier_normal_(m.weight)
                    m.bias.zero_()

    def sure_trainpipe(self, batch_size):
        return self.dataset.train().shuffled_inter_source().batch_(batch_size).tensor_()

    def marked_params(self):
        emb = [p for n, p in self.named_parameters() if "embeddings" in n]
        ids = {id(p) for p in emb}
        return [{"params": emb, "weight_decay": cfg.embedding_decay},
                {"params": [p for p in self.parameters() if id(p) not in ids], "weight_decay": cfg.weight_decay}]

    def encode(self, data):
        E = torch.stack([f.embeddings(data[f]).reshape(-1, cfg.embedding_dim) for f in self.input_fields], 1)      # [B, F, D]
        return self.fm(data, E) + self.dnn(E.flatten(1))

    def fit(self, data):
        return {"rec_loss": self.criterion(self.encode(data), data[self.Label])}

    def recommend_from_pool(self, data):
        return torch.sigmoid(self.encode(data))


class CoachForDeepFM(freerec.launcher.Coach):
    def set_optimizer(self):
        if str(self.cfg.optimizer).lower() != "adam":
            raise NotImplementedError(f"Unexpected optimizer {self.cfg.optimizer} ...")
        self.optimizer = torch.optim.Adam(self.model.marked_params(), lr=self.cfg.lr, betas=(self.cfg.adam_beta1, self.cfg.adam_beta2))

    def set_lr_scheduler(self):
        self.lr_scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="max", patience=self.cfg.eval_freq,
                                                                       **(self.cfg.get("lr_scheduler", None) or {}))

    def train_per_epoch(self, epoch):
        self.lr_scheduler.step(self._best)
        for data in self.dataloader:
            data = self.dict_to_device(data)
            loss = self.model(data)["rec_loss"]
            self.optimizer.zero_grad()
            loss.backward()
            nn.utils.clip_grad_norm_(self.model.parameters(), 10)
            self.optimizer.step()
            self.monitor(loss.item(), n=data[self.Size], reduction="mean", mode="train", pool=["LOSS"])


def main():
    try:
        dataset = getattr(freerec.data.datasets, cfg.dataset)(root=cfg.root, cfg=cfg.get("fields", None))
    except AttributeError:
        dataset = freerec.data.datasets.PredictionRecDataSet(cfg.root, cfg.dataset, tasktag=cfg.tasktag, cfg=cfg.get("fields", None))
    model = DeepFM(dataset)
    coach = CoachForDeepFM(dataset=dataset, trainpipe=model.sure_trainpipe(cfg.batch_size), validpipe=model.sure_validpipe(),
                           testpipe=model.sure_testpipe(), model=model, cfg=cfg)
    return coach.fit()


if __name__ == "__main__":
    main()
