"""Pins the CPU oracle (oracle/) against golden vectors produced by the REFERENCE classes
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import adam, criterions, deepfm, embedding, lightgcn, mf, ranking, sasrec

G = os.path.join(os.path.dirname(__file__), "golden")
torch.set_num_threads(2)


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def T(a, grad=False):
    t = torch.from_numpy(np.array(a))
    return t.requires_grad_(grad) if grad else t


# ------------------------------------------------------------------ SASRec
@pytest.mark.parametrize("loss,suffix", [("BCE", ""), ("BPR", ""), ("CE", ""), ("BCE", "_d128")])
def test_sasrec_fit_loss_and_grads(loss, suffix):
    z = load(f"sasrec_{loss.lower()}{suffix}")
    P = sasrec.params_from_npz(z, requires_grad=True)
    L = sasrec.fit(P, T(z["in/seq"]), T(z["in/pos"]), T(z["in/neg"]), loss=loss, num_blocks=int(z["cfg/num_blocks"]))
    np.testing.assert_allclose(L.item(), float(z["out/rec_loss"]), rtol=2e-6)
    L.backward()
    for k in z.files:
        if not k.startswith("grad/"):
            continue
        name = k[5:]
        g = P[name].grad
        g = torch.zeros_like(P[name]) if g is None else g
        ref = z[k]
        scale = max(np.abs(ref).max(), 1e-6)
        assert np.abs(g.numpy() - ref).max() <= 2e-5 * scale + 1e-7, name


@pytest.mark.parametrize("fixture", ["sasrec_bce", "sasrec_bce_d128"])
def test_sasrec_encode_scores_topk(fixture):
    z = load(fixture)
    P = sasrec.params_from_npz(z)
    seq = T(z["in/seq"])
    with torch.no_grad():
        u, items = sasrec.encode(P, seq, int(z["cfg/num_blocks"]))
        sc = sasrec.recommend_from_full(P, seq, int(z["cfg/num_blocks"]))
    np.testing.assert_allclose(u.numpy(), z["out/userEmbds"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(sc.numpy(), z["out/scores"], rtol=1e-4, atol=2e-5)
    # C oracle (fmaf chain + ties->lowest index) against the reference's torch.topk after scores[seen] = -1e23
    q = u[:, -1, :].numpy()
    vals, idx = ranking.score_topk(q, items.numpy(), z["in/seen_ptr"], z["in/seen_idx"], 50)
    np.testing.assert_array_equal(idx, z["out/topk_idx"])        # bit-exact top-K indices
    np.testing.assert_allclose(vals, z["out/topk_vals"], rtol=1e-4, atol=2e-5)
    dense = ranking.score_dense(q, items.numpy())
    np.testing.assert_allclose(dense, z["out/scores"], rtol=1e-4, atol=2e-5)


def test_sasrec_embed_matches_reference_frontend():
    z = load("sasrec_bce")
    x = embedding.sasrec_embed(z["param/Item.embeddings.weight"], z["param/Position.weight"], z["in/seq"])
    E, Pm, seq = T(z["param/Item.embeddings.weight"]), T(z["param/Position.weight"]), T(z["in/seq"])
    ref = (E[seq] * 8.0 + Pm[None]).masked_fill((seq == 0)[..., None], 0.0)
    np.testing.assert_array_equal(x, ref.numpy())
    # dense embedding grad == scatter_add of the upstream rows, padding row zero
    g = np.random.default_rng(0).standard_normal((8, 50, 64)).astype(np.float32)
    Eg = T(z["param/Item.embeddings.weight"], True)
    torch.nn.functional.embedding(seq, Eg, padding_idx=0).backward(T(g))
    mine = embedding.scatter_add_rows(g, z["in/seq"], 201, padding_idx=0)
    np.testing.assert_allclose(mine, Eg.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(mine, embedding.scatter_add_rows_fast(g, z["in/seq"], 201, 0))
    np.testing.assert_array_equal(mine, ranking.scatter_add_rows_c(g, z["in/seq"], 201, 0))
    np.testing.assert_array_equal(embedding.gather_rows(z["param/Item.embeddings.weight"], z["in/seq"]),
                                  ranking.gather_rows_c(z["param/Item.embeddings.weight"], z["in/seq"]))


# ------------------------------------------------------------------ MF-BPR / LightGCN
def test_mfbpr():
    z = load("mfbpr")
    U, I = T(z["param/User.embeddings.weight"], True), T(z["param/Item.embeddings.weight"], True)
    users, pos, neg = T(z["in/users"]), T(z["in/pos"]), T(z["in/neg"])
    L = mf.fit(U, I, users, pos, neg)
    np.testing.assert_allclose(L.item(), float(z["out/rec_loss"]), rtol=1e-6)
    L.backward()
    np.testing.assert_allclose(U.grad.numpy(), z["grad/User.embeddings.weight"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(I.grad.numpy(), z["grad/Item.embeddings.weight"], rtol=1e-5, atol=1e-7)
    with torch.no_grad():
        sc = mf.recommend_from_full(U, I, users)
    np.testing.assert_allclose(sc.numpy(), z["out/scores"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ranking.score_dense(U.detach().numpy()[z["in/users"][:, 0]], I.detach().numpy()),
                               z["out/scores"], rtol=1e-5, atol=1e-6)


def test_lightgcn():
    z = load("lightgcn")
    U, I = T(z["param/User.embeddings.weight"], True), T(z["param/Item.embeddings.weight"], True)
    crow, col, val = z["in/adj_crow"], z["in/adj_col"], z["in/adj_val"]
    users, pos, neg = T(z["in/users"]), T(z["in/pos"]), T(z["in/neg"])
    rec, emb = lightgcn.fit(U, I, crow, col, val, users, pos, neg, int(z["cfg/num_layers"]))
    np.testing.assert_allclose(rec.item(), float(z["out/rec_loss"]), rtol=2e-6)
    np.testing.assert_allclose(emb.item(), float(z["out/emb_loss"]), rtol=2e-6)
    (rec + float(z["cfg/weight_decay"]) * emb).backward()
    np.testing.assert_allclose(U.grad.numpy(), z["grad/User.embeddings.weight"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(I.grad.numpy(), z["grad/Item.embeddings.weight"], rtol=2e-5, atol=1e-7)
    with torch.no_grad():
        ue, ie = lightgcn.encode(U, I, crow, col, val, int(z["cfg/num_layers"]))
    np.testing.assert_allclose(ue.numpy(), z["out/userEmbds"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ie.numpy(), z["out/itemEmbds"], rtol=1e-5, atol=1e-6)


def test_lightgcn_adj_builder_matches_golden_adj():
    z = load("lightgcn")
    crow, col, val = z["in/adj_crow"], z["in/adj_col"], z["in/adj_val"]
    U, N = 30, 40
    rows = np.repeat(np.arange(U + N), np.diff(crow))
    m = rows < U
    c2, k2, v2 = lightgcn.sym_normalized_adj(U, N, rows[m], col[m] - U)
    np.testing.assert_array_equal(c2, crow)
    np.testing.assert_array_equal(k2, col)
    np.testing.assert_allclose(v2, val, rtol=1e-6)


# ------------------------------------------------------------------ DeepFM
def _deepfm_params(z, grad):
    nf = len(z["cfg/counts"])
    tables = [T(z[f"table/{i}"], grad) for i in range(nf)]
    tables_lr = [T(z[f"table_lr/{i}"], grad) for i in range(nf)]
    mlp = []
    i = 0
    while f"param/dnn.{i}.linear.weight" in z.files:
        mlp.append({k: T(z[f"param/dnn.{i}.{k}"], grad and "running" not in k)
                    for k in ("linear.weight", "linear.bias", "bn.weight", "bn.bias", "bn.running_mean", "bn.running_var")})
        i += 1
    mlp.append({"weight": T(z[f"param/dnn.{i}.weight"], grad), "bias": T(z[f"param/dnn.{i}.bias"], grad)})
    return tables, tables_lr, T(z["param/fm.lr_layer.bias"], grad), mlp


def test_pool_ranking_restatement_matches_the_reference_recommend_from_pool():
    """tests/golden/pool.npz (the reference's recommend_from_pool of SASRec / MF / LightGCN under the other fixtures' states): the oracle's
    pool scores = its full scores at the pool's columns, and the top-K rule puts the target (position 0) in front of its duplicates."""
    olg, osas = lightgcn, sasrec
    zp = np.load(os.path.join(G, "pool.npz"))
    z = np.load(os.path.join(G, "sasrec_bce.npz"))
    P = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}
    with torch.no_grad():
        u, items = osas.encode(P, torch.from_numpy(z["in/seq"]), int(z["cfg/num_blocks"]))
    sc = ranking.score_pool(u[:, -1, :].numpy(), items.numpy(), zp["sasrec/pool"])
    np.testing.assert_allclose(sc, zp["sasrec/scores"], rtol=1e-5, atol=1e-5)
    _, idx = ranking.pool_topk(sc, 21)
    for b, dup in ((0, 5), (3, 20)):                   # rows with a planted duplicate of the target: the chain gives both the same value (the
        assert sc[b, 0] == sc[b, dup]                  # reference's batched einsum differs in the last bit between them), position 0 ranks first
        pos = {int(i): r for r, i in enumerate(idx[b])}
        assert pos[0] + 1 == pos[dup]
    z = np.load(os.path.join(G, "mfbpr.npz"))
    sc = ranking.score_pool(z["param/User.embeddings.weight"][z["in/users"].reshape(-1)], z["param/Item.embeddings.weight"], zp["mfbpr/pool"])
    np.testing.assert_allclose(sc, zp["mfbpr/scores"], rtol=1e-5, atol=1e-5)
    z = np.load(os.path.join(G, "lightgcn.npz"))
    with torch.no_grad():
        ue, ie = olg.encode(torch.from_numpy(z["param/User.embeddings.weight"]), torch.from_numpy(z["param/Item.embeddings.weight"]), z["in/adj_crow"], z["in/adj_col"],
                            z["in/adj_val"], int(z["cfg/num_layers"]))
    sc = ranking.score_pool(ue.numpy()[z["in/users"].reshape(-1)], ie.numpy(), zp["lightgcn/pool"])
    np.testing.assert_allclose(sc, zp["lightgcn/scores"], rtol=1e-5, atol=1e-5)
    v, i = ranking.pool_topk(np.array([[1.0, 3.0, 3.0, 2.0]], np.float32), 6)
    assert i.tolist() == [[1, 2, 3, 0, -1, -1]] and v[0, 4] == -np.inf


def test_deepfm():
    z = load("deepfm")
    tables, tables_lr, lrb, mlp = _deepfm_params(z, True)
    x, y = T(z["in/x"]), T(z["in/labels"])
    logits = deepfm.encode(tables, tables_lr, lrb, mlp, x, training=True)
    np.testing.assert_allclose(logits.detach().numpy(), z["out/train_logits"], rtol=1e-4, atol=1e-5)
    L = criterions.bce_with_logits(logits, y)
    np.testing.assert_allclose(L.item(), float(z["out/rec_loss"]), rtol=1e-5)
    L.backward()
    for i, t in enumerate(tables):
        np.testing.assert_allclose(t.grad.numpy(), z[f"gtable/{i}"], rtol=1e-3, atol=2e-6)
        np.testing.assert_allclose(tables_lr[i].grad.numpy(), z[f"gtable_lr/{i}"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(mlp[0]["linear.weight"].grad.numpy(), z["grad/dnn.0.linear.weight"], rtol=1e-3, atol=2e-6)
    # eval pass uses the running stats as updated by the train-mode forward
    for i in range(len(mlp) - 1):
        mlp[i]["bn.running_mean"] = T(z[f"post/dnn.{i}.bn.running_mean"])
        mlp[i]["bn.running_var"] = T(z[f"post/dnn.{i}.bn.running_var"])
    with torch.no_grad():
        sc = torch.sigmoid(deepfm.encode(tables, tables_lr, lrb, mlp, x, training=False))
    np.testing.assert_allclose(sc.numpy(), z["out/eval_scores"], rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------ criteria / metrics / adam known answers
def test_criteria_known_answers():
    z = torch.zeros(7)
    assert abs(criterions.bpr_loss(z, z).item() - np.log(2)) < 1e-7          # untrained BPR = ln 2
    assert abs(criterions.bce_with_logits(z, torch.ones(7)).item() - np.log(2)) < 1e-7
    assert criterions.regularize_l2([torch.ones(3, 2)]).item() == 3.0


def test_metric_known_answers():
    rng = np.random.default_rng(0)
    B, N = 64, 300
    scores = rng.standard_normal((B, N)).astype(np.float32)
    tgt = rng.integers(0, N, B)
    targets = np.zeros((B, N), np.float32)
    targets[np.arange(B), tgt] = 1
    m = ranking.metrics_dense(scores, targets)
    np.testing.assert_array_equal(m["HITRATE@1"], m["NDCG@1"])               # one target per user
    for k in (5, 10, 20, 50):
        assert (m[f"NDCG@{k}"] <= m[f"HITRATE@{k}"] + 1e-12).all()
        assert (m[f"NDCG@{k}"] >= m[f"HITRATE@{k}"] / np.log2(k + 1) - 1e-12).all()
        np.testing.assert_allclose(m[f"RECALL@{k}"], m[f"HITRATE@{k}"])
        np.testing.assert_allclose(m[f"PRECISION@{k}"], m[f"HITRATE@{k}"] / k)
    rank = (scores > scores[np.arange(B), tgt][:, None]).sum(1)
    np.testing.assert_allclose(m["HITRATE@10"], (rank < 10).astype(float))
    np.testing.assert_allclose(m["NDCG@10"], np.where(rank < 10, 1 / np.log2(rank + 2.0), 0.0))
    np.testing.assert_allclose(m["MRR@10"], np.where(rank < 10, 1 / (rank + 1.0), 0.0))


def test_topk_ties_and_short_catalog():
    Q = np.ones((2, 4), np.float32)
    E = np.zeros((6, 4), np.float32)
    E[[1, 3, 4]] = 1.0          # three-way tie at 4.0, three-way tie at 0.0
    vals, idx = ranking.score_topk(Q, E, np.array([0, 1, 1]), np.array([3]), 5)
    np.testing.assert_array_equal(idx[1], [1, 3, 4, 0, 2])                  # ties -> lowest index
    np.testing.assert_array_equal(idx[0], [1, 4, 0, 2, 5])                  # item 3 masked for user 0
    vals, idx = ranking.score_topk(Q, E, np.array([0, 5, 5]), np.array([0, 1, 2, 3, 4]), 4)
    np.testing.assert_array_equal(idx[0], [5, 0, 1, 2])                     # K > #unmasked: masked fill by index
    assert vals[0, 1] == np.float32(-1e23)


def test_adam_matches_torch():
    rng = np.random.default_rng(1)
    p0 = rng.standard_normal((50, 8)).astype(np.float32)
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([tp], lr=5e-4, betas=(0.9, 0.999), weight_decay=1e-6)
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for step in range(1, 6):
        g = rng.standard_normal(p0.shape).astype(np.float32)
        g[::3] = 0                                       # rows with zero grad still move (dense semantics)
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        adam.adam_step(p, g, m, v, step, 5e-4, 0.9, 0.999, 1e-8, 1e-6)
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=1e-5, atol=1e-7)


def test_sparse_adam_oracle_matches_torch_sparse_adam():
    """oracle/adam.py:sparse_adam_rows (the rule re_sparse_adam_rows implements) == torch.optim.SparseAdam over several steps
    with duplicate indices (weight_decay = 0: SparseAdam has none)."""
    import torch
    from oracle import adam as oadam
    rng = np.random.default_rng(4)
    R, D = 40, 8
    W0 = rng.standard_normal((R, D)).astype(np.float32)
    emb = torch.nn.Embedding(R, D, sparse=True)
    with torch.no_grad():
        emb.weight.copy_(torch.from_numpy(W0))
    opt = torch.optim.SparseAdam(emb.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    W, m, v = W0.copy(), np.zeros_like(W0), np.zeros_like(W0)
    for step in range(1, 5):
        idx = rng.integers(0, R, 30)
        coef = rng.standard_normal((30, D)).astype(np.float32)
        opt.zero_grad()
        (emb(torch.from_numpy(idx)) * torch.from_numpy(coef)).sum().backward()
        opt.step()
        oadam.sparse_adam_rows(W, m, v, idx, coef, step, 1e-2)
        np.testing.assert_allclose(W, emb.weight.detach().numpy(), rtol=2e-5, atol=2e-6)


def test_published_benchmark_rows_satisfy_the_metric_restatements_identities():
    """The only fixtures the reference holds for freerec's metric definitions are its published result rows
    (benchmark/*/{MF-BPR,LightGCN,SASRec}.json -> tests/golden/benchmark_rows.json, made by make_benchmark_fixture.py).  With ONE
    held-out target per user (leave-one-out) the restated definitions (oracle/ranking.py) imply, for user MEANS too:
    NDCG@K <= HR@K, NDCG@K >= HR@K / log2(K + 1), both non-decreasing in K, NDCG@5 >= HR@1 (a rank-0 hit counts 1 in both);
    an untrained BPR / BCE loss is ln 2 / 2 ln 2, a trained one is below it.  Every published row must satisfy them -- and the
    oracle's metrics on synthetic scores satisfy the same identities (test_metric_known_answers), which is what pins the two together."""
    import json
    rows = json.load(open(os.path.join(G, "benchmark_rows.json")))
    assert len(rows) >= 18 and sum(len(v) for v in rows.values()) >= 90
    ks = (5, 10, 20, 50)
    for name, runs in rows.items():
        for run in runs:
            for split in ("valid", "test", "best"):
                m = run[split]
                hr = {k: m[f"HITRATE@{k}"] for k in (1,) + ks}
                nd = {k: m[f"NDCG@{k}"] for k in ks}
                for k in ks:
                    assert nd[k] <= hr[k] + 1e-12, (name, split, k)
                    assert nd[k] >= hr[k] / np.log2(k + 1) - 1e-12, (name, split, k)
                    assert nd[k] >= hr[1] - 1e-12, (name, split, k)
                seq_k = (1,) + ks
                assert all(hr[a] <= hr[b] + 1e-12 for a, b in zip(seq_k, seq_k[1:])), (name, split)
                assert all(nd[a] <= nd[b] + 1e-12 for a, b in zip(ks, ks[1:])), (name, split)
            loss = run["train"]["LOSS"]
            bound = 2 * np.log(2) if name.endswith("SASRec") else np.log(2)     # BCE(pos) + BCE(neg)  /  BPR
            assert 0.0 < loss < bound, (name, loss)


def test_sibling_oracles_reproduce_reference_fixtures():
    """oracle/siblings.py (DCN logits, SimGCL losses and tables) against the vectors made by the reference's DCN/main.py, SimGCL/main.py."""
    import torch
    from oracle import siblings
    g = np.load(os.path.join(G, "dcn.npz"))
    nf = len(g["cfg/counts"])
    T = lambda k: torch.from_numpy(g[k])  # noqa: E731
    dnn = [(T(f"param/dnn.{i}.linear.weight"), T(f"param/dnn.{i}.linear.bias"), T(f"param/dnn.{i}.bn.weight"), T(f"param/dnn.{i}.bn.bias"))
           for i in range(2)]
    cross = [(T(f"param/crossnet.{l}.weight.weight"), T(f"param/crossnet.{l}.bias")) for l in range(int(g["cfg/num_layers"]))]
    logits = siblings.dcn_logits([T(f"table/{i}") for i in range(nf)], T("in/x"), dnn, cross, (T("param/fc.weight"), T("param/fc.bias")))
    np.testing.assert_allclose(logits.numpy(), g["out/train_logits"], rtol=1e-5, atol=1e-6)
    s = np.load(os.path.join(G, "simgcl.npz"))
    S = lambda k: torch.from_numpy(s[k])  # noqa: E731
    rec, emb, ssl, ue, ie = siblings.simgcl_losses(S("param/User.embeddings.weight"), S("param/Item.embeddings.weight"), S("in/adj_crow"),
                                                   S("in/adj_col"), S("in/adj_val"), S("in/users").reshape(-1), S("in/pos").reshape(-1),
                                                   S("in/neg").reshape(-1), int(s["cfg/num_layers"]), float(s["cfg/temperature"]))
    for v, k in ((rec, "rec_loss"), (emb, "emb_loss"), (ssl, "ssl_loss")):
        assert abs(float(v) - float(s["out/" + k])) <= 1e-5 * abs(float(s["out/" + k])), k
    np.testing.assert_allclose(ue.numpy(), s["out/userEmbds"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ie.numpy(), s["out/itemEmbds"], rtol=1e-5, atol=1e-6)


def test_bert4rec_oracle_reproduces_reference_fixture():
    """oracle/siblings.py's post-norm encoder restatement against BERT4Rec/main.py's own states, loss and scores (bert4rec.npz)."""
    import torch
    from oracle import siblings
    g = np.load(os.path.join(G, "bert4rec.npz"))
    sd = {k[len("param/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}
    seq_eval = torch.from_numpy(g["in/seq_eval"])
    h = siblings.bert4rec_states(sd, seq_eval, 2, 4)
    real = (seq_eval != 0).numpy()          # (eval mode: torch's nested-tensor fast path returns zeros at padded positions)
    np.testing.assert_allclose(h.numpy()[real], g["out/states_eval"][real], rtol=1e-4, atol=2e-5)
    scores = (h[:, -1, :] @ sd["fc.weight"].t() + sd["fc.bias"])[:, 2:]
    np.testing.assert_allclose(scores.numpy(), g["out/scores"], rtol=1e-4, atol=2e-5)
    loss = siblings.bert4rec_loss(sd, torch.from_numpy(g["in/seq"]), torch.from_numpy(g["in/rnds"]), 0.3, 2, 4)
    assert abs(float(loss) - float(g["out/rec_loss"])) <= 1e-5 * abs(float(g["out/rec_loss"]))


def test_jgcf_oracle_reproduces_reference_fixture():
    import torch
    from oracle import siblings
    g = np.load(os.path.join(G, "jgcf.npz"))
    T = lambda k: torch.from_numpy(g[k])  # noqa: E731
    ue, ie = siblings.jgcf_tables(T("param/User.embeddings.weight"), T("param/Item.embeddings.weight"), T("in/adj_crow"), T("in/adj_col"),
                                  T("in/adj_val"), int(g["cfg/num_layers"]), float(g["cfg/alpha"]), float(g["cfg/beta"]),
                                  float(g["cfg/scaling_factor"]), float(g["cfg/weight4mid"]), T("param/conv.gammas"))
    np.testing.assert_allclose(ue.numpy(), g["out/userEmbds"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ie.numpy(), g["out/itemEmbds"], rtol=1e-5, atol=1e-6)
    u, p, n = (g["in/" + k].reshape(-1) for k in ("users", "pos", "neg"))
    rec = torch.nn.functional.softplus((ue[u] * ie[n]).sum(-1) - (ue[u] * ie[p]).sum(-1)).mean()
    assert abs(float(rec) - float(g["out/rec_loss"])) <= 1e-5 * abs(float(g["out/rec_loss"]))


def test_c_oracle_is_clean_under_asan_ubsan():
    """SURVEY.md section 5 (sanitizers, host only): the C restatement built with -fsanitize=address,undefined (make -C oracle asan) runs this
    file's golden checks in a child process with libasan preloaded; any report aborts the child."""
    import shutil
    import subprocess
    import sys
    if os.environ.get("RECORACLE_LIB"):
        pytest.skip("already inside the sanitizer run")
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    asan = subprocess.check_output([gcc, "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan):
        pytest.skip("libasan is not installed")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               RECORACLE_LIB=os.path.join(root, "oracle", "_build", "librecoracle_asan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "not asan", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
