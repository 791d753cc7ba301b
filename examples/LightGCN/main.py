"""LightGCN as a client of the `freerec` surface -- the recengine's own model file with the reference's structure and parameter names
(`User.embeddings.weight`, `Item.embeddings.weight`, buffer `Adj`), so checkpoints interchange with RecBoard's LightGCN/main.py.

    python examples/LightGCN/main.py --root data --dataset MyDataset           # the LightGCN engine after the adoption probe
    python examples/LightGCN/main.py ... --engine module                       # the torch code below

Arithmetic (LightGCN/main.py:77-108,131-172 of the reference): X0 = [U; I]; X_{l+1} = Adj X_l; out = mean(X_0 .. X_L); BPR on `out` rows;
emb = (|U0[u]|^2 + |I0[i+]|^2 + |I0[i-]|^2) / 2 / B on the RAW rows; the Coach differentiates rec + cfg.weight_decay * emb and its
optimizer carries NO weight decay."""
import freerec
import torch
import torch.nn as nn

freerec.declare(version="1.0.1")

cfg = freerec.parser.Parser()
cfg.add_argument("--embedding-dim", type=int, default=64)
cfg.add_argument("--num-layers", type=int, default=3)
cfg.set_defaults(description="LightGCN", root="../../data", dataset="Yelp2018_10100_LOU", epochs=1000, batch_size=2048,
                 optimizer="adam", lr=1e-3, weight_decay=1e-4, seed=1)
cfg.compile()


class LightGCN(freerec.models.GenRecArch):
    def __init__(self, dataset):
        super().__init__(dataset)
        self.num_layers = cfg.num_layers
        for field in (self.User, self.Item):
            field.add_module("embeddings", nn.Embedding(field.count, cfg.embedding_dim))
        self.register_buffer("Adj", self.dataset.train().to_normalized_adj(normalization="sym"))
        self.criterion = freerec.criterions.BPRLoss(reduction="mean")
        with torch.no_grad():
            for field in (self.User, self.Item):
                field.embeddings.weight.normal_(0.0, 1e-4)

    def sure_trainpipe(self, batch_size):
        return (self.dataset.train().choiced_user_ids_source().gen_train_sampling_pos_()
                .gen_train_sampling_neg_(num_negatives=1).batch_(batch_size).tensor_())

    def encode(self):
        x = torch.cat((self.User.embeddings.weight, self.Item.embeddings.weight), 0)
        layers = [x]
        for _ in range(self.num_layers):
            x = self.Adj @ x
            layers.append(x)
        out = torch.stack(layers, 0).mean(0)
        return out[:self.User.count], out[self.User.count:]

    def fit(self, data):
        U, I = self.encode()
        users, pos, neg = data[self.User], data[self.IPos], data[self.INeg]
        u = U[users]
        rec = self.criterion((u * I[pos]).sum(-1), (u * I[neg]).sum(-1))
        raw = [self.User.embeddings(users), self.Item.embeddings(pos), self.Item.embeddings(neg)]
        return {"rec_loss": rec, "emb_loss": self.criterion.regularize(raw, rtype="l2") / len(users)}

    def reset_ranking_buffers(self):
        U, I = self.encode()
        self.ranking_buffer = {self.User: U.detach().clone(), self.Item: I.detach().clone()}

    def recommend_from_full(self, data):
        return self.ranking_buffer[self.User][data[self.User]].squeeze(1) @ self.ranking_buffer[self.Item].t()

    def recommend_from_pool(self, data):
        u = self.ranking_buffer[self.User][data[self.User]]
        return (u * self.ranking_buffer[self.Item][data[self.IUnseen]]).sum(-1)


class CoachForLightGCN(freerec.launcher.Coach):
    def set_optimizer(self):
        # the L2 term is in the loss (below); the optimizer itself decays nothing
        if str(self.cfg.optimizer).lower() != "adam":
            raise NotImplementedError(f"Unexpected optimizer {self.cfg.optimizer} ...")
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=self.cfg.lr,
                                          betas=(self.cfg.optim_first_moment_decay, self.cfg.optim_second_moment_decay))

    def train_per_epoch(self, epoch):
        for data in self.dataloader:
            data = self.dict_to_device(data)
            out = self.model(data)
            loss = out["rec_loss"] + self.cfg.weight_decay * out["emb_loss"]
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()
            self.monitor(loss.item(), n=len(data[self.User]), reduction="mean", mode="train", pool=["LOSS"])


def main():
    try:
        dataset = getattr(freerec.data.datasets, cfg.dataset)(root=cfg.root)
    except AttributeError:
        dataset = freerec.data.datasets.RecDataSet(cfg.root, cfg.dataset, tasktag=cfg.tasktag)
    model = LightGCN(dataset)
    coach = CoachForLightGCN(dataset=dataset, trainpipe=model.sure_trainpipe(cfg.batch_size), validpipe=model.sure_validpipe(cfg.ranking),
                             testpipe=model.sure_testpipe(cfg.ranking), model=model, cfg=cfg)
    return coach.fit()


if __name__ == "__main__":
    main()
