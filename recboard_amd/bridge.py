"""Engine routing for the `freerec`-compatible surface (freerec/launcher.py): recognise a model the engine has a fused step for and
drive it through that step with the MODULE's parameters living in the engine's arena.

SASRec (SASRec/main.py:53-228): `Item.embeddings` [N + 1, D], `Position` [S, D], `attnLNs / attnLayers (nn.MultiheadAttention, one
head) / fwdLNs / fwdLayers (conv1, conv2: Conv1d k = 1)`, `lastLN`, `criterion` BCE / BPR -- the state-dict names the engine's arena uses
are the reference's, so the module's state dict loads as is, and afterwards every module parameter IS the arena view of that name
(`p.data = view`): the script's own `encode / recommend_from_full`, `state_dict()` and checkpoints read what the fused step trains."""
import torch

from .sasrec import SASRecEngine


class SASRecAdapter:
    def __init__(self, coach, module, loss):
        cfg = coach.cfg
        D = module.Item.embeddings.weight.shape[1]
        S = module.Position.weight.shape[0]
        L = len(module.attnLayers)
        p = float(module.embdDropout.p) if hasattr(module, "embdDropout") else float(cfg.get("dropout_rate", 0.0))
        betas = (cfg.get("beta1", cfg.get("adam_beta1", 0.9)), cfg.get("beta2", cfg.get("adam_beta2", 0.999)))
        self.module = module
        self.eng = SASRecEngine(module.Item.count, S, D, L, dropout_rate=p, loss=loss, lr=float(cfg.lr), weight_decay=float(cfg.weight_decay),
                                betas=betas, device=coach.device, seed=int(cfg.get("seed", 1)))
        sd = {k: v for k, v in module.state_dict().items() if k in self.eng.params}
        missing = [k for k in self.eng.params if k not in sd]
        if missing:
            raise KeyError(f"not a SASRec state dict: {missing[:3]}")
        self.eng.load_state_dict(sd)
        named = dict(module.named_parameters())
        for k, view in self.eng.params.items():          # the module's parameters become the arena's views
            named[k].data = view.detach().view(named[k].shape)

    def train_epoch(self, coach, epoch):
        from .coach import _lookahead
        eng = self.eng.train()
        n = 0
        eng.begin_loss_accumulation()       # (every step's loss x its batch size is folded into one device word by the next step's stage launch)

        def batches():
            for data in coach.dataloader:
                if "Sample" in data:      # the device sampler's ticket (freerec pipe -> .to_(device)): a step's launches sample the batch
                    yield data["Sample"]
                else:
                    seq, pos, neg = (data[f].to(coach.device, non_blocking=True) for f in (coach.ISeq, coach.IPos, coach.INeg))
                    yield (seq, pos.reshape(seq.shape), neg.reshape(seq.shape))

        # one batch ahead: every step is told the next batch (or ticket), which its tail launch prepares (SASRecEngine._train_step_graph_tail)
        for cur, nxt in _lookahead(batches()):
            if isinstance(cur, tuple):
                loss = eng.train_step_graph(*cur, next_batch=nxt if isinstance(nxt, tuple) else None)
                bsz = cur[0].shape[0]
            else:
                loss = eng.train_step_graph_sampled(cur, next_ticket=nxt if nxt is not None and not isinstance(nxt, tuple) else None)
                bsz = len(cur)
            n += bsz
        tot = eng.end_loss_accumulation().reshape(())
        eng.check_handover()
        coach.monitor(float(tot / max(n, 1)), n=max(n, 1), reduction="mean", mode="train", pool=["LOSS"])   # (one host read per epoch)

    def reset_ranking_buffers(self):
        self.eng.eval()
        self.eng.reset_ranking_buffers()

    def recommend_topk(self, coach, data, seen_ptr, seen_idx, K):
        return self.eng.recommend_topk(data[coach.ISeq].to(coach.device), seen_ptr, seen_idx, K)

    def optimizer_state(self):
        return self.eng.arena.adam_state_dict(self.eng.lr, self.eng.betas, self.eng.wd)

    def load_optimizer_state(self, sd):
        self.eng.arena.load_adam_state_dict(sd)


def attach(coach):
    """-> an adapter when `coach.model` is a model the engine runs fused (and the optimizer is Adam), else None."""
    m = coach.get_res_sys_arch()
    cfg = coach.cfg
    need = ("Item", "Position", "attnLNs", "attnLayers", "fwdLNs", "fwdLayers", "lastLN", "criterion")
    if not all(hasattr(m, a) for a in need) or str(cfg.get("optimizer", "adam")).lower() != "adam":
        return None
    try:
        crit = type(m.criterion).__name__
        loss = {"BCELoss4Logits": "BCE", "BPRLoss": "BPR"}.get(crit)
        mha = m.attnLayers[0]
        D = m.Item.embeddings.weight.shape[1]
        ok = (loss is not None and D in (64, 128) and mha.num_heads == 1 and m.Position.weight.shape[0] <= 64 and len(m.attnLayers) <= 4
              and hasattr(m.fwdLayers[0], "conv1") and m.Item.embeddings.weight.shape[0] == m.Item.count + 1)
        if not ok:
            return None
        return SASRecAdapter(coach, m, loss)
    except Exception:  # noqa: BLE001  (anything unexpected about the module: leave it to its own torch code)
        return None
