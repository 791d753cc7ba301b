"""GPU: sibling baselines on the torch custom ops (recboard_amd.siblings) against golden vectors made by importing the reference's
DCN/main.py and SimGCL/main.py (tests/golden/make_golden.py): forward values, losses and every parameter gradient."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def test_dcn_matches_reference_forward_and_gradients():
    from recboard_amd.siblings import DCN
    g = np.load(os.path.join(GOLD, "dcn.npz"))
    counts = g["cfg/counts"].tolist()
    m = DCN(counts, embedding_dim=10, hidden_dims=(32, 24), num_layers=int(g["cfg/num_layers"]), batch_norm=True)
    sd = m.state_dict()
    with torch.no_grad():
        m.embeddings.weight.copy_(torch.cat([_t(g[f"table/{i}"]) for i in range(len(counts))]))
        for k in sd:
            if k.startswith(("dnn.", "crossnet.", "fc.")):
                sd[k].copy_(_t(g["param/" + k]).view(sd[k].shape))
    x, labels = _t(g["in/x"]), _t(g["in/labels"])
    m.train()
    logits = m.encode(x)
    torch.testing.assert_close(logits, _t(g["out/train_logits"]), rtol=1e-4, atol=1e-6)
    loss = m.criterion(logits, labels.to(torch.float32))
    assert abs(float(loss) - float(g["out/rec_loss"])) <= 1e-5 * abs(float(g["out/rec_loss"]))
    loss.backward()
    for k, p in m.named_parameters():
        if k == "embeddings.weight":
            ref = torch.cat([_t(g[f"gtable/{i}"]) for i in range(len(counts))])
        else:
            ref = _t(g["grad/" + k]).view(p.shape)
        torch.testing.assert_close(p.grad, ref, rtol=2e-4, atol=2e-7, msg=k)
    # the train-mode forward updated the BatchNorm running statistics as the reference's did
    for i in range(2):
        torch.testing.assert_close(m.dnn[i].bn.running_mean, _t(g[f"post/dnn.{i}.bn.running_mean"]), rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(m.dnn[i].bn.running_var, _t(g[f"post/dnn.{i}.bn.running_var"]), rtol=1e-5, atol=1e-7)
    m.eval()
    with torch.no_grad():
        torch.testing.assert_close(m.recommend_from_pool(x), _t(g["out/eval_scores"]), rtol=1e-4, atol=1e-6)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        DCN(counts, device="cpu").encode(torch.zeros(2, len(counts), dtype=torch.long))


def test_simgcl_matches_reference_losses_and_gradients():
    from recboard_amd.siblings import SimGCL
    g = np.load(os.path.join(GOLD, "simgcl.npz"))
    U, N = g["param/User.embeddings.weight"].shape[0], g["param/Item.embeddings.weight"].shape[0]
    adj = (_t(g["in/adj_crow"]), _t(g["in/adj_col"]), _t(g["in/adj_val"]))
    m = SimGCL(U, N, adj, embedding_dim=g["param/User.embeddings.weight"].shape[1], num_layers=int(g["cfg/num_layers"]), eps=0.0,
               temperature=float(g["cfg/temperature"]))
    with torch.no_grad():
        m.user.weight.copy_(_t(g["param/User.embeddings.weight"]))
        m.item.weight.copy_(_t(g["param/Item.embeddings.weight"]))
    users, pos, neg = (_t(g["in/" + k]).reshape(-1) for k in ("users", "pos", "neg"))
    m.train()
    losses = m.fit(users, pos, neg)
    for k in ("rec_loss", "emb_loss", "ssl_loss"):
        assert abs(float(losses[k]) - float(g["out/" + k])) <= 2e-5 * abs(float(g["out/" + k])), k
    (losses["rec_loss"] + losses["emb_loss"] + losses["ssl_loss"]).backward()
    torch.testing.assert_close(m.user.weight.grad, _t(g["grad/User.embeddings.weight"]), rtol=2e-4, atol=2e-6)
    torch.testing.assert_close(m.item.weight.grad, _t(g["grad/Item.embeddings.weight"]), rtol=2e-4, atol=2e-6)
    m.eval()
    ue, ie = m.encode()
    torch.testing.assert_close(ue, _t(g["out/userEmbds"]), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ie, _t(g["out/itemEmbds"]), rtol=1e-5, atol=1e-6)
    m.reset_ranking_buffers()
    torch.testing.assert_close(m.recommend_from_full(users), _t(g["out/scores"]), rtol=1e-5, atol=1e-6)


def test_linear_on_engine_gemm_matches_torch_linear_both_ways():
    from recboard_amd import nn as rnn
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(37, 50, device="cuda", generator=g, requires_grad=True)
    lin = rnn.Linear(50, 24, device="cuda")
    ref = torch.nn.Linear(50, 24, device="cuda")
    with torch.no_grad():
        ref.weight.copy_(lin.weight); ref.bias.copy_(lin.bias.normal_(generator=g))
    y = lin(x)
    y.pow(2).sum().backward()
    gx, gw, gb = x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()
    x.grad = None
    yr = ref(x)
    yr.pow(2).sum().backward()
    torch.testing.assert_close(y, yr, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gx, x.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gw, ref.weight.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gb, ref.bias.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("loss", ["BCE", "BPR"])
def test_gru4rec_matches_reference_loss_gradients_and_scores(loss):
    from recboard_amd.siblings import GRU4Rec
    g = np.load(os.path.join(GOLD, f"gru4rec_{loss.lower()}.npz"))
    N, H = int(g["cfg/N"]), int(g["cfg/hidden"])
    m = GRU4Rec(N, embedding_dim=64, hidden_size=H, num_blocks=1, emb_dropout_rate=0.0, hidden_dropout_rate=0.0, loss=loss)
    with torch.no_grad():
        m.item.weight.copy_(_t(g["param/Item.embeddings.weight"]))
        m.gru.weight_ih_l0.copy_(_t(g["param/gru.weight_ih_l0"])); m.gru.weight_hh_l0.copy_(_t(g["param/gru.weight_hh_l0"]))
        m.dense.weight.copy_(_t(g["param/dense.weight"])); m.dense.bias.copy_(_t(g["param/dense.bias"]))
    seq, pos, neg = _t(g["in/seq"]), _t(g["in/pos"]).reshape(-1), _t(g["in/neg"]).reshape(-1)
    m.train()
    out = m.fit(seq, pos, neg)["rec_loss"]
    assert abs(float(out.detach()) - float(g["out/rec_loss"])) <= 2e-5 * abs(float(g["out/rec_loss"]))
    out.backward()
    for name, p in (("Item.embeddings.weight", m.item.weight), ("gru.weight_ih_l0", m.gru.weight_ih_l0), ("gru.weight_hh_l0", m.gru.weight_hh_l0),
                    ("dense.weight", m.dense.weight), ("dense.bias", m.dense.bias)):
        torch.testing.assert_close(p.grad, _t(g["grad/" + name]).view(p.shape), rtol=3e-4, atol=2e-6, msg=name)
    assert float(m.item.weight.grad[0].abs().max()) == 0.0      # the padding row takes no gradient
    m.eval()
    with torch.no_grad():
        torch.testing.assert_close(m.recommend_from_full(seq), _t(g["out/scores"]), rtol=1e-4, atol=1e-5)


def test_ngcf_matches_reference_losses_gradients_and_scores():
    """NGCF on the LEFT-normalised adjacency with self loops (not symmetric): nn.spmm's backward runs on the transposed CSR."""
    from recboard_amd.siblings import NGCF
    g = np.load(os.path.join(GOLD, "ngcf.npz"))
    U, N = g["param/User.embeddings.weight"].shape[0], g["param/Item.embeddings.weight"].shape[0]
    adj = (_t(g["in/adj_crow"]), _t(g["in/adj_col"]), _t(g["in/adj_val"]))
    m = NGCF(U, N, adj, embedding_dim=g["param/User.embeddings.weight"].shape[1], num_layers=int(g["cfg/num_layers"]), dropout_rate=0.0)
    sd = m.state_dict()
    with torch.no_grad():
        m.user.weight.copy_(_t(g["param/User.embeddings.weight"])); m.item.weight.copy_(_t(g["param/Item.embeddings.weight"]))
        for k in sd:
            if k.startswith("convs."):
                sd[k].copy_(_t(g["param/" + k]).view(sd[k].shape))
    users, pos, neg = (_t(g["in/" + k]).reshape(-1) for k in ("users", "pos", "neg"))
    m.train()
    losses = m.fit(users, pos, neg)
    for k in ("rec_loss", "emb_loss"):
        assert abs(float(losses[k].detach()) - float(g["out/" + k])) <= 2e-5 * abs(float(g["out/" + k])), k
    (losses["rec_loss"] + losses["emb_loss"]).backward()
    torch.testing.assert_close(m.user.weight.grad, _t(g["grad/User.embeddings.weight"]), rtol=3e-4, atol=2e-6)
    torch.testing.assert_close(m.item.weight.grad, _t(g["grad/Item.embeddings.weight"]), rtol=3e-4, atol=2e-6)
    for k, p in m.named_parameters():
        if k.startswith("convs."):
            torch.testing.assert_close(p.grad, _t(g["grad/" + k]).view(p.shape), rtol=3e-4, atol=2e-6, msg=k)
    m.eval()
    with torch.no_grad():
        ue, ie = m.encode()
        torch.testing.assert_close(ue, _t(g["out/userEmbds"]), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(ie, _t(g["out/itemEmbds"]), rtol=1e-5, atol=1e-6)
        m.reset_ranking_buffers()
        torch.testing.assert_close(m.recommend_from_full(users), _t(g["out/scores"]), rtol=1e-4, atol=1e-5)


def test_bert4rec_matches_reference_loss_gradients_and_scores():
    """The masked-position rows and the `fc` projection run on the engine ops; the mask draw of the reference run is replayed."""
    from recboard_amd.siblings import BERT4Rec
    g = np.load(os.path.join(GOLD, "bert4rec.npz"))
    N, S = int(g["cfg/N"]), int(g["cfg/maxlen"])
    m = BERT4Rec(N, maxlen=S, embedding_dim=64, num_heads=4, num_blocks=2, mask_ratio=0.3, dropout_rate=0.0)
    names = {"item.weight": "Item.embeddings.weight"}
    sd = m.state_dict()
    with torch.no_grad():
        for k in sd:
            sd[k].copy_(_t(g["param/" + names.get(k, k)]).view(sd[k].shape))
    seq = _t(g["in/seq"])
    m.train()
    loss = m.fit(seq, rnds=_t(g["in/rnds"]))["rec_loss"]
    assert abs(float(loss.detach()) - float(g["out/rec_loss"])) <= 2e-5 * abs(float(g["out/rec_loss"]))
    loss.backward()
    for k, p in m.named_parameters():
        torch.testing.assert_close(p.grad, _t(g["grad/" + names.get(k, k)]).view(p.shape), rtol=5e-4, atol=2e-6, msg=k)
    assert float(m.item.weight.grad[0].abs().max()) == 0.0      # the padding row takes no gradient
    m.eval()
    with torch.no_grad():
        torch.testing.assert_close(m.recommend_from_full(_t(g["in/seq_eval"])), _t(g["out/scores"]), rtol=1e-4, atol=2e-5)


def test_jgcf_matches_reference_losses_gradients_and_scores():
    from recboard_amd.siblings import JGCF
    g = np.load(os.path.join(GOLD, "jgcf.npz"))
    U, N = g["param/User.embeddings.weight"].shape[0], g["param/Item.embeddings.weight"].shape[0]
    adj = (_t(g["in/adj_crow"]), _t(g["in/adj_col"]), _t(g["in/adj_val"]))
    m = JGCF(U, N, adj, embedding_dim=g["param/User.embeddings.weight"].shape[1], num_layers=int(g["cfg/num_layers"]),
             scaling_factor=float(g["cfg/scaling_factor"]), alpha=float(g["cfg/alpha"]), beta=float(g["cfg/beta"]),
             weight4mid=float(g["cfg/weight4mid"]))
    with torch.no_grad():
        m.user.weight.copy_(_t(g["param/User.embeddings.weight"])); m.item.weight.copy_(_t(g["param/Item.embeddings.weight"]))
        m.gammas.copy_(_t(g["param/conv.gammas"]))
    users, pos, neg = (_t(g["in/" + k]).reshape(-1) for k in ("users", "pos", "neg"))
    m.train()
    losses = m.fit(users, pos, neg)
    for k in ("rec_loss", "emb_loss"):
        assert abs(float(losses[k].detach()) - float(g["out/" + k])) <= 2e-5 * abs(float(g["out/" + k])), k
    (losses["rec_loss"] + losses["emb_loss"]).backward()
    torch.testing.assert_close(m.user.weight.grad, _t(g["grad/User.embeddings.weight"]), rtol=3e-4, atol=2e-6)
    torch.testing.assert_close(m.item.weight.grad, _t(g["grad/Item.embeddings.weight"]), rtol=3e-4, atol=2e-6)
    m.eval()
    with torch.no_grad():
        ue, ie = m.encode()
        torch.testing.assert_close(ue, _t(g["out/userEmbds"]), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(ie, _t(g["out/itemEmbds"]), rtol=1e-5, atol=1e-6)
        m.reset_ranking_buffers()
        torch.testing.assert_close(m.recommend_from_full(users), _t(g["out/scores"]), rtol=1e-4, atol=1e-5)


def test_graphed_step_replays_the_eager_step():
    """nn.GraphedStep: the captured DCN step (engine ops + BatchNorm + capturable Adam) follows the eager one, batch after batch."""
    from recboard_amd import nn as rnn
    from recboard_amd.siblings import DCN
    counts = [50, 20, 7, 3]
    g = torch.Generator().manual_seed(2)
    xs = [torch.stack([torch.randint(0, c, (64,), generator=g) for c in counts], 1).cuda() for _ in range(4)]
    ys = [(torch.rand(64, generator=g) < 0.4).float().cuda() for _ in range(4)]

    def make():
        torch.manual_seed(11)
        m = DCN(counts, embedding_dim=8, hidden_dims=(32, 16), num_layers=2, batch_norm=True)
        with torch.no_grad():
            m.embeddings.weight.mul_(1e3)
        return m.train(), None

    (a, _), (b, _) = make(), make()
    oa = torch.optim.Adam(a.parameters(), lr=1e-2, capturable=True)
    ob = torch.optim.Adam(b.parameters(), lr=1e-2, capturable=True)
    step = rnn.GraphedStep(b, lambda x, y: b.fit(x, y)["rec_loss"], ob, (xs[0], ys[0]))
    for k, p in a.named_parameters():                  # capture left the parameters where they were
        torch.testing.assert_close(dict(b.named_parameters())[k], p, rtol=0, atol=0)
    for bn in (m for m in b.modules() if isinstance(m, torch.nn.BatchNorm1d)):     # (the warm-up moved the running statistics: reset both)
        bn.reset_running_stats()
    for x, y in zip(xs, ys):
        oa.zero_grad(set_to_none=True)
        la = a.fit(x, y)["rec_loss"]
        la.backward()
        oa.step()
        lb = step(x, y)
        assert abs(float(la.detach()) - float(lb)) <= 1e-5 * abs(float(la.detach()))
    for k, p in a.named_parameters():
        torch.testing.assert_close(dict(b.named_parameters())[k], p, rtol=1e-4, atol=1e-6, msg=k)


def _load_by_name(m, g, names):
    """Copy the reference's state-dict entries into the sibling's parameters (names: sibling name -> reference name)."""
    with torch.no_grad():
        for k, p in m.named_parameters():
            p.copy_(_t(g["param/" + names.get(k, k)]).view(p.shape))


def _check_grads(m, g, names, rtol=5e-4, atol=2e-6):
    for k, p in m.named_parameters():
        torch.testing.assert_close(p.grad, _t(g["grad/" + names.get(k, k)]).view(p.shape), rtol=rtol, atol=atol, msg=k)


def test_gcn_matches_reference_loss_gradients_and_scores():
    from recboard_amd.siblings import GCN
    g = np.load(os.path.join(GOLD, "gcn.npz"))
    U, N = g["param/User.embeddings.weight"].shape[0], g["param/Item.embeddings.weight"].shape[0]
    adj = (_t(g["in/adj_crow"]), _t(g["in/adj_col"]), _t(g["in/adj_val"]))
    m = GCN(U, N, adj, embedding_dim=g["param/User.embeddings.weight"].shape[1], num_layers=int(g["cfg/num_layers"]))
    names = {"user.weight": "User.embeddings.weight", "item.weight": "Item.embeddings.weight"}
    _load_by_name(m, g, names)
    users, pos, neg = (_t(g["in/" + k]).reshape(-1) for k in ("users", "pos", "neg"))
    m.train()
    loss = m.fit(users, pos, neg)["rec_loss"]
    assert abs(float(loss.detach()) - float(g["out/rec_loss"])) <= 2e-5 * abs(float(g["out/rec_loss"]))
    loss.backward()
    _check_grads(m, g, names)
    m.eval()
    with torch.no_grad():
        ue, ie = m.encode()
        torch.testing.assert_close(ue, _t(g["out/userEmbds"]), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(ie, _t(g["out/itemEmbds"]), rtol=1e-4, atol=1e-6)
        m.reset_ranking_buffers()
        torch.testing.assert_close(m.recommend_from_full(users), _t(g["out/scores"]), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("which", ["stamp_bce", "stamp_ce", "narm", "fmlprec_bpr", "bsarec_ce"])
def test_last_item_siblings_match_reference_loss_gradients_and_scores(which):
    """STAMP, NARM, FMLP-Rec, BSARec on the engine ops against vectors made from STAMP/main.py, NARM/main.py, FMLP-Rec/main.py + modules.py,
    BSARec/main.py + modules.py."""
    from recboard_amd import siblings as sib
    g = np.load(os.path.join(GOLD, which + ".npz"))
    N, S = int(g["cfg/N"]), int(g["cfg/maxlen"])
    names = {"item.weight": "Item.embeddings.weight"}
    if which.startswith("stamp"):
        m = sib.STAMP(N, 64, 64, loss=which[6:].upper())
    elif which == "narm":
        m = sib.NARM(N, 64, 48, 1, emb_dropout_rate=0.0, hidden_dropout_rate=0.0, ct_dropout_rate=0.0)
    elif which == "bsarec_ce":      # (the sibling mirrors the reference's module tree: only the item table is named differently)
        m = sib.BSARec(N, maxlen=S, embedding_dim=64, num_heads=2, num_blocks=2, c=5, alpha=0.7, hidden_dropout_rate=0.0, attn_dropout_rate=0.0, loss="CE")
    else:
        m = sib.FMLPRec(N, maxlen=S, embedding_dim=64, num_blocks=2, hidden_dropout_rate=0.0, loss="BPR")
        for l in range(2):
            pre = f"itemEncoder.layer.{l}."
            names.update({f"blocks.{l}.complex_weight": pre + "filterlayer.complex_weight",
                          f"blocks.{l}.filter_norm.weight": pre + "filterlayer.LayerNorm.weight", f"blocks.{l}.filter_norm.bias": pre + "filterlayer.LayerNorm.bias",
                          f"blocks.{l}.dense_1.weight": pre + "intermediate.dense_1.weight", f"blocks.{l}.dense_1.bias": pre + "intermediate.dense_1.bias",
                          f"blocks.{l}.dense_2.weight": pre + "intermediate.dense_2.weight", f"blocks.{l}.dense_2.bias": pre + "intermediate.dense_2.bias",
                          f"blocks.{l}.out_norm.weight": pre + "intermediate.LayerNorm.weight", f"blocks.{l}.out_norm.bias": pre + "intermediate.LayerNorm.bias"})
    _load_by_name(m, g, names)
    seq, pos, neg = _t(g["in/seq"]), _t(g["in/pos"]).reshape(-1), _t(g["in/neg"]).reshape(-1)
    m.train()
    loss = m.fit(seq, pos, neg)["rec_loss"]
    assert abs(float(loss.detach()) - float(g["out/rec_loss"])) <= 2e-5 * abs(float(g["out/rec_loss"]))
    loss.backward()
    _check_grads(m, g, names)
    assert float(m.item.weight.grad[0].abs().max()) == 0.0      # the padding row takes no gradient
    m.eval()
    with torch.no_grad():
        torch.testing.assert_close(m.recommend_from_full(seq), _t(g["out/scores"]), rtol=1e-4, atol=1e-5)


def _toy_interactions(U, N, per_user, seed):
    rng = np.random.default_rng(seed)
    hist = [sorted(rng.choice(N, per_user, replace=False).tolist()) for _ in range(U)]
    return hist


@pytest.mark.parametrize("graph", [False, True])
def test_coach_runs_a_graph_sibling_end_to_end(graph):
    """Coach(kind="module"): GCN on the custom-op surface trained by the Coach's epoch loop (eager and as one hipGraph per step), evaluated
    through the fused score + seen-mask + top-K; the metrics equal the dense restatement (scores -> -1e23 at seen -> torch.topk)."""
    from recboard_amd import graph as rgraph
    from recboard_amd.coach import Coach
    from recboard_amd.siblings import GCN
    U, N, B = 60, 90, 32
    hist = _toy_interactions(U, N, 7, 3)
    train = [h[:-1] for h in hist]
    target = [[h[-1]] for h in hist]
    eu = np.concatenate([[u] * len(t) for u, t in enumerate(train)]); ei = np.concatenate(train)
    crow, col, val = (torch.from_numpy(a) for a in rgraph.to_normalized_adj(U, N, eu, ei, "sym"))
    torch.manual_seed(0)
    m = GCN(U, N, (crow, col, val), embedding_dim=64, num_layers=2)
    with torch.no_grad():
        m.user.weight.normal_(std=0.1); m.item.weight.normal_(std=0.1)
    rng = np.random.default_rng(5)
    trainpipe = []
    for _ in range(6):
        users = rng.integers(0, U, B)
        pos = np.array([rng.choice(train[u]) for u in users])
        neg = rng.integers(0, N, B)
        trainpipe.append({"User": torch.from_numpy(users), "IPos": torch.from_numpy(pos), "INeg": torch.from_numpy(neg)})
    validpipe = [{"User": torch.arange(u0, min(U, u0 + 25)), "ISeen": train[u0:u0 + 25], "IUnseen": target[u0:u0 + 25]} for u0 in range(0, U, 25)]
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, capturable=graph)
    coach = Coach(m, trainpipe, validpipe, monitors=("LOSS", "HitRate@10", "NDCG@10"), which4best="NDCG@10", eval_freq=1, kind="module",
                  optimizer=opt, fit_keys=("User", "IPos", "INeg"), graph=graph)
    out = coach.fit(3)
    losses = [h["train"]["LOSS"] for h in out["history"] if "train" in h]      # (the last record holds the final evaluations)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    # dense restatement of the last evaluation
    m.reset_ranking_buffers()
    ue, ie = m.ranking_buffer
    scores = (ue @ ie.t()).cpu().numpy()
    hr = nd = 0.0
    for u in range(U):
        s = scores[u].copy(); s[train[u]] = -1e23
        top = np.argsort(-s, kind="stable")[:10]
        hit = np.nonzero(top == target[u][0])[0]
        if hit.size:
            hr += 1; nd += 1.0 / np.log2(hit[0] + 2)
    got = out["history"][-1]["valid"]
    assert abs(got["HITRATE@10"] - hr / U) < 1e-6 and abs(got["NDCG@10"] - nd / U) < 1e-5


def test_coach_runs_a_sequence_sibling():
    """Coach(kind="module") with STAMP: fit on (ISeq, IPos, INeg), evaluation on ISeq through `recommend_topk`."""
    from recboard_amd.coach import Coach
    from recboard_amd.siblings import STAMP
    N, B, S = 80, 16, 12
    rng = np.random.default_rng(1)
    torch.manual_seed(1)
    m = STAMP(N, 64, 64, loss="BPR")
    with torch.no_grad():
        m.item.weight.mul_(50.0); m.item.weight[0].zero_()

    def seqs(n):
        x = np.zeros((n, S), np.int64)
        for b in range(n):
            L = rng.integers(2, S)
            x[b, S - L:] = rng.integers(1, N + 1, L)
        return x
    trainpipe = [{"ISeq": torch.from_numpy(seqs(B)), "IPos": torch.from_numpy(rng.integers(0, N, (B, 1))), "INeg": torch.from_numpy(rng.integers(0, N, (B, 1)))}
                 for _ in range(4)]
    ev = seqs(20)
    validpipe = [{"ISeq": torch.from_numpy(ev), "ISeen": [sorted(set((r[r > 0] - 1).tolist())) for r in ev], "IUnseen": [[int(rng.integers(0, N))] for _ in ev]}]
    coach = Coach(m, trainpipe, validpipe, monitors=("LOSS", "NDCG@10"), eval_freq=1, kind="module",
                  optimizer=torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True), fit_keys=("ISeq", "IPos", "INeg"), graph=True)
    out = coach.fit(2)
    assert np.isfinite(out["history"][-2]["train"]["LOSS"]) and 0.0 <= out["history"][-1]["valid"]["NDCG@10"] <= 1.0
    # the fused top-K of the evaluation = torch.topk on the masked dense scores
    x = torch.from_numpy(ev).cuda()
    sp, si = coach_csr = __import__("recboard_amd.evaluate", fromlist=["ragged_to_csr"]).ragged_to_csr(validpipe[0]["ISeen"], "cuda")
    _, idx = m.recommend_topk(x, sp, si, 10)
    with torch.no_grad():
        dense = m.recommend_from_full(x).clone()
    for b, seen in enumerate(validpipe[0]["ISeen"]):
        dense[b, seen] = -1e23
    assert torch.equal(idx, torch.topk(dense, 10, dim=1).indices)


def test_sgl_matches_reference_on_recorded_edge_dropout():
    """SGL: the two sampled subgraphs are re-weightings of the full graph's CSR pattern; the reference run's uniform draws are replayed."""
    from recboard_amd.siblings import SGL
    g = np.load(os.path.join(GOLD, "sgl.npz"))
    U, N = g["param/User.embeddings.weight"].shape[0], g["param/Item.embeddings.weight"].shape[0]
    m = SGL(U, N, (g["in/edges"][0], g["in/edges"][1]), embedding_dim=g["param/User.embeddings.weight"].shape[1], num_layers=int(g["cfg/num_layers"]),
            aug_type="ed", ssl_drop_rate=float(g["cfg/ssl_drop_rate"]), temperature=float(g["cfg/temperature"]))
    with torch.no_grad():
        m.user.weight.copy_(_t(g["param/User.embeddings.weight"])); m.item.weight.copy_(_t(g["param/Item.embeddings.weight"]))
    m.resample(rnds=(_t(g["in/rnd1"]), _t(g["in/rnd2"])))
    # the first view's adjacency is the reference's sampled one
    sub = torch.sparse_csr_tensor(m.crow, m.col, m.sub[0], size=(U + N, U + N)).to_dense()
    torch.testing.assert_close(sub, _t(g["out/sub_adj_dense"]), rtol=1e-5, atol=1e-7)
    assert float((m.sub[0] == 0).float().mean()) > 0.1          # (edges were dropped)
    users, pos, neg = (_t(g["in/" + k]).reshape(-1) for k in ("users", "pos", "neg"))
    m.train()
    losses = m.fit(users, pos, neg)
    for k in ("rec_loss", "emb_loss", "ssl_loss"):
        assert abs(float(losses[k].detach()) - float(g["out/" + k])) <= 2e-5 * abs(float(g["out/" + k])), k
    (losses["rec_loss"] + losses["emb_loss"] + losses["ssl_loss"]).backward()
    torch.testing.assert_close(m.user.weight.grad, _t(g["grad/User.embeddings.weight"]), rtol=3e-4, atol=2e-6)
    torch.testing.assert_close(m.item.weight.grad, _t(g["grad/Item.embeddings.weight"]), rtol=3e-4, atol=2e-6)
    m.eval()
    with torch.no_grad():
        ue, ie = m.encode()
        torch.testing.assert_close(ue, _t(g["out/userEmbds"]), rtol=1e-5, atol=1e-6)
        m.reset_ranking_buffers()
        torch.testing.assert_close(m.recommend_from_full(users), _t(g["out/scores"]), rtol=1e-5, atol=1e-6)
    # node dropout and random walk draw their own subgraphs
    for aug in ("nd", "rw"):
        m2 = SGL(U, N, (g["in/edges"][0], g["in/edges"][1]), embedding_dim=64, num_layers=2, aug_type=aug, ssl_drop_rate=0.2)
        out = m2.fit(users, pos, neg)
        assert all(torch.isfinite(v) for v in out.values())
        assert (m2.sub[0] is m2.sub[1]) == (aug == "nd")


def _right_padded(rng, B, S, N, offset=1):
    seq = np.zeros((B, S), np.int64)
    for b in range(B):
        n = int(rng.integers(1, S - 3))                       # (at least three all-pad columns at the right: shrink_pads has something to drop)
        seq[b, :n] = rng.integers(0, N, n) + offset
    return seq


@pytest.mark.parametrize("name", ["GRU4Rec", "NARM"])
def test_static_shapes_give_the_reference_form_results(name):
    """static_shapes (no shrink_pads: no data-dependent shape, no host sync) computes the same loss and gradients as the reference form."""
    from recboard_amd import siblings
    N, B, S = 300, 16, 12
    rng = np.random.default_rng(5)
    seq = torch.from_numpy(_right_padded(rng, B, S, N)).cuda()
    pos, neg = (torch.from_numpy(rng.integers(0, N, (B, 1))).cuda() for _ in range(2))
    kw = dict(emb_dropout_rate=0.0, hidden_dropout_rate=0.0)
    if name == "NARM":
        kw["ct_dropout_rate"] = 0.0
    torch.manual_seed(3)
    m = getattr(siblings, name)(N, embedding_dim=32, hidden_size=48, **kw).train()
    out = []
    for static in (False, True):
        m.static_shapes = static
        m.zero_grad(set_to_none=True)
        loss = m.fit(seq, pos, neg)["rec_loss"]
        loss.backward()
        out.append((loss.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    torch.testing.assert_close(out[1][0], out[0][0], rtol=1e-5, atol=1e-7)
    for k, g in out[0][1].items():
        torch.testing.assert_close(out[1][1][k], g, rtol=1e-4, atol=1e-6, msg=k)


def test_bert4rec_static_shapes_loss_equals_masked_indexing():
    from recboard_amd.siblings import BERT4Rec
    N, B, S = 200, 12, 20
    rng = np.random.default_rng(9)
    seq = np.zeros((B, S), np.int64)
    for b in range(B):
        n = int(rng.integers(2, S))
        seq[b, S - n:] = rng.integers(0, N, n) + 2
    seq = torch.from_numpy(seq).cuda()
    rnds = torch.from_numpy(rng.random((B, S))).float().cuda()
    torch.manual_seed(4)
    m = BERT4Rec(N, maxlen=S, embedding_dim=32, num_heads=2, num_blocks=1, dropout_rate=0.0).train()
    res = []
    for static in (False, True):
        m.static_shapes = static
        m.zero_grad(set_to_none=True)
        loss = m.fit(seq, rnds)["rec_loss"]
        loss.backward()
        res.append((loss.detach().clone(), m.fc.weight.grad.clone(), m.item.weight.grad.clone()))
    torch.testing.assert_close(res[1][0], res[0][0], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(res[1][1], res[0][1], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(res[1][2], res[0][2], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("name", ["GRU4Rec", "NARM", "BERT4Rec"])
def test_sibling_steps_capture_into_a_graph(name):
    """With static shapes the whole training step of the three models replays from one hipGraph (nn.GraphedStep through the Coach)."""
    from recboard_amd import siblings
    from recboard_amd.coach import Coach
    N, B, S = 300, 32, 12
    rng = np.random.default_rng(6)
    torch.manual_seed(8)
    if name == "BERT4Rec":
        m = siblings.BERT4Rec(N, maxlen=S, embedding_dim=32, num_heads=2, num_blocks=1, dropout_rate=0.1).train()
        pipe = [{"ISeq": torch.from_numpy(np.where(rng.random((B, S)) < 0.3, 0, rng.integers(0, N, (B, S)) + 2)).cuda()} for _ in range(6)]
        keys = ("ISeq",)
    else:
        kw = dict(emb_dropout_rate=0.1, hidden_dropout_rate=0.0)
        m = getattr(siblings, name)(N, embedding_dim=32, hidden_size=48, **kw).train()
        pipe = [{"ISeq": torch.from_numpy(_right_padded(rng, B, S, N)).cuda(), "IPos": torch.from_numpy(rng.integers(0, N, (B, 1))).cuda(),
                 "INeg": torch.from_numpy(rng.integers(0, N, (B, 1))).cuda()} for _ in range(6)]
        keys = ("ISeq", "IPos", "INeg")
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    coach = Coach(m, pipe, monitors=["LOSS"], kind="module", optimizer=torch.optim.Adam(m.parameters(), lr=1e-2, capturable=True), fit_keys=keys,
                  graph=True)
    l0 = coach.train_per_epoch(0)["LOSS"]
    for _ in range(4):
        l1 = coach.train_per_epoch(1)["LOSS"]
    assert m.static_shapes and len(coach._graphed) == 1
    assert np.isfinite(l0) and np.isfinite(l1) and l1 < l0                                # (it trains: 30 replayed steps on 6 batches)
    assert any(not torch.equal(before[k], p.detach()) for k, p in m.named_parameters())
