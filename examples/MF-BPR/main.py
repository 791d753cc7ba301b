"""MF-BPR as a client of the `freerec` surface -- the recengine's own model file, laid out the way RecBoard's model scripts are (module-level
`cfg`, a `GenRecArch` subclass with `sure_trainpipe / encode / fit / reset_ranking_buffers / recommend_from_*`, a `Coach` subclass with the
step loop, `main()`), with the reference's parameter names (`User.embeddings.weight`, `Item.embeddings.weight`), so checkpoints interchange
with RecBoard's MF-BPR/main.py.

    python examples/MF-BPR/main.py --root data --dataset MyDataset             # the MF engine after the adoption probe (cfg.engine = auto)
    python examples/MF-BPR/main.py ... --engine module                         # the torch code below, op by op

Arithmetic (MF-BPR/main.py:78-104 of the reference): score(u, i) = <U[u], I[i]>; loss = mean softplus(score(u, i-) - score(u, i+)); tables
~ N(0, 1e-4^2); evaluation on tables cloned once per pass."""
import freerec
import torch
import torch.nn as nn

freerec.declare(version="1.0.1")

cfg = freerec.parser.Parser()
cfg.add_argument("--embedding-dim", type=int, default=64)
cfg.set_defaults(description="MF-BPR", root="../../data", dataset="Amazon2014Beauty_550_LOU", epochs=1000, batch_size=2048,
                 optimizer="adam", lr=1e-3, weight_decay=1e-4, seed=1)
cfg.compile()


class MF(freerec.models.GenRecArch):
    def __init__(self, dataset):
        super().__init__(dataset)
        for field in (self.User, self.Item):
            field.add_module("embeddings", nn.Embedding(field.count, cfg.embedding_dim))
        self.criterion = freerec.criterions.BPRLoss(reduction="mean")
        with torch.no_grad():
            for field in (self.User, self.Item):
                field.embeddings.weight.normal_(0.0, 1e-4)

    def sure_trainpipe(self, batch_size):
        return (self.dataset.train().choiced_user_ids_source().gen_train_sampling_pos_()
                .gen_train_sampling_neg_(num_negatives=1).batch_(batch_size).tensor_())

    def encode(self):
        return self.User.embeddings.weight, self.Item.embeddings.weight

    def fit(self, data):
        U, I = self.encode()
        u = U[data[self.User]]                                        # [B, 1, D]
        pos = (u * I[data[self.IPos]]).sum(-1)                        # [B, 1]
        neg = (u * I[data[self.INeg]]).sum(-1)                        # [B, K]
        return {"rec_loss": self.criterion(pos, neg)}

    def reset_ranking_buffers(self):
        U, I = self.encode()
        self.ranking_buffer = {self.User: U.detach().clone(), self.Item: I.detach().clone()}

    def recommend_from_full(self, data):
        return self.ranking_buffer[self.User][data[self.User]].squeeze(1) @ self.ranking_buffer[self.Item].t()

    def recommend_from_pool(self, data):
        u = self.ranking_buffer[self.User][data[self.User]]
        return (u * self.ranking_buffer[self.Item][data[self.IUnseen]]).sum(-1)


class CoachForMF(freerec.launcher.Coach):
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
    model = MF(dataset)
    coach = CoachForMF(dataset=dataset, trainpipe=model.sure_trainpipe(cfg.batch_size), validpipe=model.sure_validpipe(cfg.ranking),
                       testpipe=model.sure_testpipe(cfg.ranking), model=model, cfg=cfg)
    return coach.fit()


if __name__ == "__main__":
    main()
