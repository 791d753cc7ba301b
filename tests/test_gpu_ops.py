"""GPU parity tests: HIP kernels (through the C ABI) vs the CPU oracle and the committed golden vectors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from recboard_amd import ops as _ops
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def zipf_ids(rng, n, N, a=1.0):
    w = 1.0 / np.arange(1, N + 1) ** a
    return rng.choice(N, size=n, p=w / w.sum()).astype(np.int64)


# ------------------------------------------------------------------------------------------------ gather
@pytest.mark.parametrize("R,D,n", [(201, 64, 400), (1000, 128, 777), (37, 10, 50), (9, 1, 33), (5000, 32, 4096), (300, 256, 100)])
def test_gather_rows_bit_exact(ops, R, D, n):
    from oracle import embedding
    rng = np.random.default_rng(R + D)
    W = rng.standard_normal((R, D)).astype(np.float32)
    idx = rng.integers(0, R, n)
    idx[: min(n, 5)] = idx[0]  # duplicates
    out = ops.gather_rows(dev(W), dev(idx)).cpu().numpy()
    np.testing.assert_array_equal(out, embedding.gather_rows(W, idx))


def test_gather_rows_shapes_empty_and_oob(ops):
    W = torch.randn(50, 64, device="cuda")
    assert ops.gather_rows(W, torch.zeros((0,), dtype=torch.long, device="cuda")).shape == (0, 64)
    idx = torch.tensor([[0, 49], [-1, 50]], device="cuda")
    out = ops.gather_rows(W, idx)
    assert out.shape == (2, 2, 64)
    torch.testing.assert_close(out[0], W[[0, 49]], rtol=0, atol=0)
    assert (out[1] == 0).all()  # out-of-range indices never fault: zero rows


def test_gather_rows_large_zipf_checksum(ops):
    # BASELINE-sized gather (SASRec/Beauty: 512*50 lookups into [12102, 64]) via a size-independent property
    rng = np.random.default_rng(1)
    W = rng.standard_normal((12102, 64)).astype(np.float32)
    idx = zipf_ids(rng, 512 * 50, 12102)
    out = ops.gather_rows(dev(W), dev(idx))
    ref = torch.from_numpy(W)[torch.from_numpy(idx)]
    assert torch.equal(out.cpu(), ref)


def test_sasrec_embed_matches_oracle_and_golden(ops):
    from oracle import embedding, rng as orng
    z = np.load(os.path.join(G, "sasrec_bce.npz"))
    E, P, seq = z["param/Item.embeddings.weight"], z["param/Position.weight"], z["in/seq"]
    out = ops.sasrec_embed(dev(E), dev(P), dev(seq), 8.0).cpu().numpy()
    np.testing.assert_array_equal(out, embedding.sasrec_embed(E, P, seq))
    # engine dropout mask == its restatement in oracle/rng.py
    p, seed = 0.5, 1234
    outd = ops.sasrec_embed(dev(E), dev(P), dev(seq), 8.0, p, seed).cpu().numpy()
    keep = orng.keep_mask(seed, orng.STREAM_EMBED, seq.shape + (64,), p)
    ref = embedding.sasrec_embed(E, P, seq) * keep * np.float32(1.0 / (1.0 - p))
    np.testing.assert_allclose(outd, ref, rtol=1e-6, atol=0)
    assert 0.45 < keep.mean() < 0.55


# ------------------------------------------------------------------------------------------------ scatter-add
@pytest.mark.parametrize("R,D,n,pad", [(201, 64, 400, 0), (50, 64, 5000, -1), (1000, 128, 3000, -1), (37, 10, 500, -1),
                                        (9, 1, 200, -1), (70000, 64, 2048, -1), (12102, 64, 25600, 0)])
def test_scatter_add_rows_vs_oracle(ops, R, D, n, pad):
    from oracle import ranking
    rng = np.random.default_rng(R * 7 + n)
    g = rng.standard_normal((n, D)).astype(np.float32)
    idx = zipf_ids(rng, n, R)  # heavy collisions on the low ids
    if pad >= 0:
        idx[rng.integers(0, n, n // 10)] = pad
    out = ops.scatter_add_rows(dev(g), dev(idx), R, pad, 1.0).cpu().numpy()
    ref = ranking.scatter_add_rows_c(g, idx, R, pad)
    scale = np.abs(ref).max() + 1e-6
    assert np.abs(out - ref).max() <= 2e-5 * scale  # fp32, different (but fixed) association for long runs
    if pad >= 0:
        assert (out[pad] == 0).all()
    untouched = np.setdiff1d(np.arange(R), idx)
    assert (out[untouched] == 0).all()


def test_scatter_add_rows_deterministic_and_scaled(ops):
    rng = np.random.default_rng(5)
    n, D, R = 30000, 64, 500
    g, idx = dev(rng.standard_normal((n, D)).astype(np.float32)), dev(zipf_ids(rng, n, R))
    a = ops.scatter_add_rows(g, idx, R, -1, 8.0)
    b = ops.scatter_add_rows(g, idx, R, -1, 8.0)
    assert torch.equal(a, b)  # bitwise reproducible
    c = ops.scatter_add_rows(g, idx, R, -1, 1.0)
    assert torch.equal(a, c * 8.0)  # power-of-two scale commutes with fp32 rounding
    # linearity: scatter(g1 + g2) ~= scatter(g1) + scatter(g2)
    g2 = torch.randn_like(g)
    lhs = ops.scatter_add_rows(g + g2, idx, R)
    rhs = ops.scatter_add_rows(g, idx, R) + ops.scatter_add_rows(g2, idx, R)
    torch.testing.assert_close(lhs, rhs, rtol=1e-4, atol=1e-3)


def test_scatter_add_rows_edge_cases(ops):
    g = torch.randn(3, 64, device="cuda")
    out = ops.scatter_add_rows(g[:0], torch.zeros((0,), dtype=torch.long, device="cuda"), 10)
    assert out.shape == (10, 64) and (out == 0).all()
    idx = torch.tensor([7, 7, 7], device="cuda")
    out = ops.scatter_add_rows(g, idx, 10)
    torch.testing.assert_close(out[7], (g[0] + g[1]) + g[2], rtol=0, atol=0)
    idx = torch.tensor([-5, 10, 3], device="cuda")  # out of range -> dropped, never a fault
    out = ops.scatter_add_rows(g, idx, 10)
    expected = torch.zeros_like(out)
    expected[3] = g[2]
    assert torch.equal(out, expected)


def test_embedding_grad_matches_golden(ops):
    """scatter_add == the reference's dense embedding grad for the SASRec front end (padding row zero)."""
    z = np.load(os.path.join(G, "sasrec_bce.npz"))
    seq = z["in/seq"]
    g = np.random.default_rng(0).standard_normal(seq.shape + (64,)).astype(np.float32)
    E = torch.from_numpy(z["param/Item.embeddings.weight"]).requires_grad_(True)
    (torch.nn.functional.embedding(torch.from_numpy(seq), E, padding_idx=0) * 8.0).backward(torch.from_numpy(g))
    out = ops.scatter_add_rows(dev(g), dev(seq), 201, 0, 8.0).cpu()
    torch.testing.assert_close(out, E.grad, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------ pair losses
@pytest.mark.parametrize("kind", ["BCE", "BPR"])
def test_pair_loss_fwd_bwd_vs_oracle(ops, kind):
    from oracle import criterions
    rng = np.random.default_rng(3)
    B, S, D, N = 16, 50, 64, 300
    U = torch.from_numpy(rng.standard_normal((B, S, D)).astype(np.float32) * 0.3).requires_grad_(True)
    E = torch.from_numpy(rng.standard_normal((N + 1, D)).astype(np.float32) * 0.3).requires_grad_(True)
    valid = torch.from_numpy(rng.random((B, S)) < 0.3)
    pos = torch.from_numpy(rng.integers(0, N, (B, S)))
    neg = torch.from_numpy(rng.integers(0, N, (B, S)))
    u = U[valid]
    pl = (u * E[1:][pos[valid]]).sum(-1)
    nl = (u * E[1:][neg[valid]]).sum(-1)
    if kind == "BCE":
        ref = criterions.bce_with_logits(pl, torch.ones_like(pl)) + criterions.bce_with_logits(nl, torch.zeros_like(nl))
    else:
        ref = criterions.bpr_loss(pl, nl)
    (ref * 1.7).backward()

    k = ops.LOSS_BCE if kind == "BCE" else ops.LOSS_BPR
    Ud, Ed = U.detach().cuda().view(-1, D), E.detach().cuda()
    vd, pd, nd = valid.cuda().to(torch.uint8).view(-1), pos.cuda().view(-1), neg.cuda().view(-1)
    loss, logits, count = ops.pair_loss_fwd(Ud, Ed, pd, nd, vd, k, e_off=1)
    assert int(count) == int(valid.sum())
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-6)
    dU, gp, gn = ops.pair_loss_bwd(Ud, Ed, pd, nd, vd, k, logits, count, torch.tensor([1.7], device="cuda"), e_off=1)
    torch.testing.assert_close(dU.cpu().view(B, S, D), U.grad, rtol=1e-4, atol=1e-7)
    rows_p = torch.where(vd.bool(), pd + 1, torch.zeros_like(pd))
    rows_n = torch.where(vd.bool(), nd + 1, torch.zeros_like(nd))
    dE = ops.scatter_add_rows(torch.cat([gp, gn]), torch.cat([rows_p, rows_n]), N + 1, 0)
    torch.testing.assert_close(dE.cpu(), E.grad, rtol=1e-4, atol=1e-6)
    assert (dU.view(B, S, D)[~valid.cuda()] == 0).all()


def test_bpr_triplet_matches_golden(ops):
    z = np.load(os.path.join(G, "mfbpr.npz"))
    Ut, It = dev(z["param/User.embeddings.weight"]), dev(z["param/Item.embeddings.weight"])
    users, pos, neg = (dev(z[k]).view(-1) for k in ("in/users", "in/pos", "in/neg"))
    loss, logits = ops.bpr_triplet_fwd(Ut, It, users, pos, neg)
    np.testing.assert_allclose(loss.item(), float(z["out/rec_loss"]), rtol=2e-6)
    gu, gp, gn = ops.bpr_triplet_bwd(Ut, It, users, pos, neg, logits, None)
    dU = ops.scatter_add_rows(gu, users, Ut.shape[0])
    dI = ops.scatter_add_rows(torch.cat([gp, gn]), torch.cat([pos, neg]), It.shape[0])
    np.testing.assert_allclose(dU.cpu().numpy(), z["grad/User.embeddings.weight"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(dI.cpu().numpy(), z["grad/Item.embeddings.weight"], rtol=1e-4, atol=1e-7)


def test_bpr_untrained_is_ln2(ops):
    # known-answer: std 1e-4 init (MF-BPR/main.py:55) => loss ~ ln 2
    g = torch.Generator(device="cuda").manual_seed(1)
    Ut = torch.randn(22363, 64, device="cuda", generator=g) * 1e-4
    It = torch.randn(12101, 64, device="cuda", generator=g) * 1e-4
    users = torch.randint(0, 22363, (2048,), device="cuda", generator=g)
    pos = torch.randint(0, 12101, (2048,), device="cuda", generator=g)
    neg = torch.randint(0, 12101, (2048,), device="cuda", generator=g)
    loss, _ = ops.bpr_triplet_fwd(Ut, It, users, pos, neg)
    assert abs(loss.item() - np.log(2)) < 1e-5


# ------------------------------------------------------------------------------------------------ scoring
@pytest.mark.parametrize("B,N,D", [(8, 200, 64), (130, 1000, 64), (33, 517, 128), (5, 64, 32), (300, 12101, 64)])
def test_score_dense_bit_exact_vs_fmaf_chain(ops, B, N, D):
    from oracle import ranking
    rng = np.random.default_rng(B + N)
    Q = rng.standard_normal((B, D)).astype(np.float32)
    E = rng.standard_normal((N, D)).astype(np.float32)
    out = ops.score_dense(dev(Q), dev(E)).cpu().numpy()
    np.testing.assert_array_equal(out, ranking.score_dense(Q, E))  # k-ordered fmaf chain, bit for bit


def _seen(rng, B, N, mean):
    lists = [np.unique(rng.integers(0, N, rng.integers(0, 2 * mean + 1))) for _ in range(B)]
    ptr = np.zeros(B + 1, np.int64)
    ptr[1:] = np.cumsum([len(x) for x in lists])
    return ptr, (np.concatenate(lists) if ptr[-1] else np.zeros(0, np.int64)).astype(np.int64)


@pytest.mark.parametrize("B,N,D,K", [(8, 200, 64, 50), (130, 1000, 64, 10), (33, 517, 128, 50), (5, 64, 32, 64),
                                      (260, 12101, 64, 50), (3, 70, 64, 1)])
def test_score_topk_bit_exact_vs_oracle(ops, B, N, D, K):
    from oracle import ranking
    rng = np.random.default_rng(B * 3 + N)
    Q = rng.standard_normal((B, D)).astype(np.float32)
    E = rng.standard_normal((N, D)).astype(np.float32)
    sp, si = _seen(rng, B, N, 9)
    vals, idx = ops.score_topk(dev(Q), dev(E), dev(sp), dev(si), K)
    rv, ri = ranking.score_topk(Q, E, sp, si, K)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)  # bit-exact top-K indices
    np.testing.assert_array_equal(vals.cpu().numpy(), rv)
    # retain_seen (no mask)
    vals, idx = ops.score_topk(dev(Q), dev(E), None, None, K)
    rv, ri = ranking.score_topk(Q, E, None, None, K)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)


def _seen_clustered(rng, B, N, heavy_every=7):
    """Seen lists that stress the cursor: runs of consecutive ids (several seen ids inside the same 16 items), a few heavy
    users (hundreds of ids: the two-id window is used up inside a stage), users with none, ids at both ends of the catalog."""
    lists = []
    for b in range(B):
        if b % 5 == 0:
            ids = np.zeros(0, np.int64)
        elif b % heavy_every == 0:
            ids = rng.choice(N, size=min(N - 1, int(rng.integers(100, 400))), replace=False)
        else:
            starts = rng.integers(0, N, rng.integers(1, 4))
            ids = np.concatenate([np.arange(s0, min(N, s0 + rng.integers(1, 9))) for s0 in starts])
            ids = np.concatenate([ids, [0, N - 1]]) if b % 3 == 0 else ids
        lists.append(np.unique(ids).astype(np.int64))
    ptr = np.zeros(B + 1, np.int64)
    ptr[1:] = np.cumsum([len(x) for x in lists])
    return ptr, np.concatenate(lists).astype(np.int64)


@pytest.mark.parametrize("B,N,K,ties,D", [(2100, 1500, 50, False, 64), (2050, 1237, 50, True, 64), (2048, 777, 10, True, 64),
                                          (2300, 1000, 32, False, 64), (2048, 1111, 51, True, 64), (2200, 901, 52, False, 64),
                                          (2048, 65, 50, True, 64), (2100, 1300, 50, False, 128), (2048, 999, 50, True, 128),
                                          (2060, 700, 20, False, 128)])
def test_score_topk_register_list_kernel_bit_exact(ops, B, N, K, ties, D):
    """The variant that serves B >= ~2000 users (per-lane sorted register lists, shared bounds, after-the-fact voiding of seen
    ids): exact parity with the oracle on catalogs small enough to check every user -- K = 50 (its own instantiation), the
    generic K <= 16 / 32 / 52 instantiations, D = 128 (one workgroup per CU), an odd K (scalar output path), catalog sizes that are not multiples of the
    32-item tile, clustered / heavy seen lists, and integer-valued embeddings that make most scores tie."""
    from oracle import ranking
    rng = np.random.default_rng(B + N + K)
    if ties:
        Q = rng.integers(-1, 2, (B, D)).astype(np.float32)
        E = rng.integers(-1, 2, (N, D)).astype(np.float32)
    else:
        Q = rng.standard_normal((B, D)).astype(np.float32)
        E = rng.standard_normal((N, D)).astype(np.float32)
    sp, si = _seen_clustered(rng, B, N)
    vals, idx = ops.score_topk(dev(Q), dev(E), dev(sp), dev(si), K)
    rv, ri = ranking.score_topk(Q, E, sp, si, K)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    np.testing.assert_array_equal(vals.cpu().numpy(), rv)
    vals2, idx2 = ops.score_topk(dev(Q), dev(E), dev(sp), dev(si), K)       # and again: no dependence on workgroup timing
    assert torch.equal(idx, idx2) and torch.equal(vals, vals2)
    vals, idx = ops.score_topk(dev(Q), dev(E), None, None, K)               # retain_seen
    rv, ri = ranking.score_topk(Q, E, None, None, K)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)


def test_score_topk_ties_and_masked_fill(ops):
    from oracle import ranking
    D = 64
    Q = np.ones((2, D), np.float32)
    E = np.zeros((200, D), np.float32)
    E[[1, 3, 4, 150]] = 1.0                         # 4-way tie on top, 196-way tie at 0
    sp, si = np.array([0, 1, 1]), np.array([3])
    vals, idx = ops.score_topk(dev(Q), dev(E), dev(sp), dev(si), 6)
    rv, ri = ranking.score_topk(Q, E, sp, si, 6)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)  # ties -> lowest index
    assert idx[1].tolist() == [1, 3, 4, 150, 0, 2]
    # K > #unmasked: the tail is the masked items in ascending order with value -1e23
    E2 = np.random.default_rng(0).standard_normal((10, D)).astype(np.float32)
    sp, si = np.array([0, 8, 8]), np.arange(8)
    vals, idx = ops.score_topk(dev(Q), dev(E2), dev(sp), dev(si), 5)
    rv, ri = ranking.score_topk(Q, E2, sp, si, 5)
    np.testing.assert_array_equal(idx.cpu().numpy(), ri)
    np.testing.assert_array_equal(vals.cpu().numpy(), rv)
    assert vals[0, 2].item() == np.float32(-1e23)


def test_score_topk_matches_reference_golden(ops):
    z = np.load(os.path.join(G, "sasrec_bce.npz"))
    q = z["out/userEmbds"][:, -1, :]
    E = z["param/Item.embeddings.weight"][1:]
    vals, idx = ops.score_topk(dev(q), dev(E), dev(z["in/seen_ptr"]), dev(z["in/seen_idx"]), 50)
    np.testing.assert_array_equal(idx.cpu().numpy(), z["out/topk_idx"])  # the reference's torch.topk indices
    np.testing.assert_allclose(vals.cpu().numpy(), z["out/topk_vals"], rtol=1e-4, atol=2e-5)
    dense = ops.score_dense(dev(q), dev(E)).cpu().numpy()
    np.testing.assert_allclose(dense, z["out/scores"], rtol=1e-4, atol=2e-5)


def test_score_topk_full_beauty_properties(ops):
    """BASELINE size (22 363 users x 12 101 items): checked through size-independent properties + a sampled oracle."""
    from oracle import ranking
    rng = np.random.default_rng(11)
    B, N, D, K = 22363, 12101, 64, 50
    Q = rng.standard_normal((B, D)).astype(np.float32)
    E = rng.standard_normal((N, D)).astype(np.float32)
    sp, si = _seen(rng, B, N, 9)
    vals, idx = ops.score_topk(dev(Q), dev(E), dev(sp), dev(si), K)
    v, i = vals.cpu().numpy(), idx.cpu().numpy()
    for _ in range(3):   # workgroups share per-user bounds through global memory: the result must not depend on their timing
        v2, i2 = ops.score_topk(dev(Q), dev(E), dev(sp), dev(si), K)
        assert np.array_equal(i2.cpu().numpy(), i) and np.array_equal(v2.cpu().numpy(), v)
    assert (np.diff(v, axis=1) <= 0).all()                                  # sorted
    assert (np.sort(i, axis=1)[:, 1:] != np.sort(i, axis=1)[:, :-1]).all()  # no duplicates
    rows = rng.integers(0, B, 200)                                          # values are the exact scores of the ids
    chk = np.einsum("bd,bkd->bk", Q[rows].astype(np.float64), E[i[rows]].astype(np.float64))
    np.testing.assert_allclose(v[rows], chk, rtol=1e-4, atol=1e-4)
    for b in rows[:50]:                                                     # none of the seen items
        assert not np.isin(i[b], si[sp[b]:sp[b + 1]]).any()
    sub = np.concatenate([np.arange(0, 300), np.arange(B - 300, B)])        # exact parity on 600 users
    rv, ri = ranking.score_topk(Q[sub], E, np.concatenate([[0], np.cumsum(np.diff(sp)[sub])]),
                                np.concatenate([si[sp[b]:sp[b + 1]] for b in sub]), K)
    np.testing.assert_array_equal(i[sub], ri)
    np.testing.assert_array_equal(v[sub], rv)


# ------------------------------------------------------------------------------------------------ adam
def test_adam_step_matches_oracle_and_torch(ops):
    from oracle import adam
    rng = np.random.default_rng(2)
    n = 12102 * 64 + 3
    p0 = rng.standard_normal(n).astype(np.float32)
    p, m, v = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pr, mr, vr = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    for step in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        g[::3] = 0
        ops.adam_step(p, dev(g), m, v, step, 5e-4, 0.9, 0.999, 1e-8, 1e-6)
        adam.adam_step(pr, g, mr, vr, step, 5e-4, 0.9, 0.999, 1e-8, 1e-6)
    np.testing.assert_allclose(p.cpu().numpy(), pr, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(v.cpu().numpy(), vr, rtol=2e-6, atol=1e-12)


# ------------------------------------------------------------------------------------------------ ranking metrics
def test_rank_metrics_vs_oracle(ops):
    from oracle import ranking
    rng = np.random.default_rng(8)
    B, N, K = 333, 500, 50
    scores = rng.standard_normal((B, N)).astype(np.float32)
    ntg = rng.integers(1, 4, B)
    tl = [np.sort(rng.choice(N, k, replace=False)) for k in ntg]
    ptr = np.zeros(B + 1, np.int64)
    ptr[1:] = np.cumsum(ntg)
    tidx = np.concatenate(tl).astype(np.int64)
    order = np.argsort(-scores, axis=1, kind="stable")[:, :K].astype(np.int64)
    ks = (1, 5, 10, 20, 50)
    ref = ranking.metrics_from_topk(order, ptr, tidx, ks)
    per_user, sums = ops.rank_metrics(dev(order), dev(ptr), dev(tidx), ks)
    for ik, k in enumerate(ks):
        for j, name in enumerate(ops.METRIC_NAMES):
            np.testing.assert_allclose(per_user[:, ik, j].cpu().numpy(), ref[f"{name}@{k}"], rtol=1e-5, atol=1e-6, err_msg=f"{name}@{k}")
            np.testing.assert_allclose(sums[ik, j].item(), ref[f"{name}@{k}"].sum(), rtol=1e-5)


def test_evaluator_matches_reference_golden_pipeline(ops):
    """Coach.evaluate contract end to end on the SASRec golden: scores -> seen mask -> top-50 -> HR/NDCG."""
    from oracle import ranking
    from recboard_amd.evaluate import RankingEvaluator
    z = np.load(os.path.join(G, "sasrec_bce.npz"))
    q = z["out/userEmbds"][:, -1, :]
    E = z["param/Item.embeddings.weight"][1:]
    tgt = np.arange(8) * 7 % 200                     # one held-out target per user (leave-one-out)
    tp, ti = np.arange(9, dtype=np.int64), tgt.astype(np.int64)
    ev = RankingEvaluator(["LOSS", "HitRate@1", "HitRate@10", "NDCG@10", "NDCG@50"])
    _, idx = ops.score_topk(dev(q), dev(E), dev(z["in/seen_ptr"]), dev(z["in/seen_idx"]), ev.kmax)
    ev.update(idx, dev(tp), dev(ti))
    got = ev.compute()
    s = z["out/scores"].copy()
    for b in range(8):
        s[b, z["in/seen_idx"][z["in/seen_ptr"][b]:z["in/seen_ptr"][b + 1]]] = -1e23
    tg = np.zeros_like(s)
    tg[np.arange(8), tgt] = 1
    ref = ranking.metrics_dense(s, tg, (1, 10, 50))
    for k in got:
        np.testing.assert_allclose(got[k], ref[k].mean(), rtol=1e-5, atol=1e-7, err_msg=k)


def test_scatter_plan_apply_equals_scatter_add_rows(ops):
    """re_scatter_plan + re_scatter_apply (the index / data halves as separate entry points) == re_scatter_add_rows, bitwise."""
    g = torch.Generator().manual_seed(3)
    n, D, R = 5000, 64, 700
    idx = torch.randint(0, R, (n,), generator=g).cuda()
    rows = torch.randn(n, D, generator=g).cuda()
    ref = ops.scatter_add_rows(rows, idx, R, padding_idx=0)
    from recboard_amd import lib
    ws = torch.empty(lib.load().re_scatter_add_rows_workspace_bytes(n, D, R), dtype=torch.uint8, device="cuda")
    out = torch.full((R, D), 7.0, device="cuda")
    ops.scatter_plan(idx, D, R, ws, padding_idx=0, zero=out)
    ops.scatter_apply(rows, R, out, ws, accumulate=True)
    assert torch.equal(out, ref)
    out2 = torch.full((R, D), 7.0, device="cuda")
    ops.scatter_plan(idx, D, R, ws, padding_idx=0)
    ops.scatter_apply(rows, R, out2, ws, accumulate=False)
    assert torch.equal(out2, ref)


@pytest.mark.parametrize("kind", [0, 1])
def test_pair_loss_fwd_bwd_equals_separate_kernels(ops, kind):
    """re_pair_loss_fwd_bwd (one pass, M from a device word) == re_pair_loss_fwd + re_pair_loss_bwd, bitwise."""
    g = torch.Generator().manual_seed(11)
    n, D, R = 3000, 64, 500
    U = torch.randn(n, D, generator=g).cuda()
    E = torch.randn(R + 1, D, generator=g).cuda()
    pos = torch.randint(0, R, (n,), generator=g).cuda()
    neg = torch.randint(0, R, (n,), generator=g).cuda()
    valid = (torch.rand(n, generator=g) < 0.4).to(torch.uint8).cuda()
    loss, logits, count = ops.pair_loss_fwd(U, E, pos, neg, valid, kind, e_off=1)
    dU, gp, gn = ops.pair_loss_bwd(U, E, pos, neg, valid, kind, logits, count, None, e_off=1)
    cnt = valid.sum(dtype=torch.int32).reshape(1)
    loss2, dU2, gp2, gn2 = ops.pair_loss_fwd_bwd(U, E, pos, neg, valid, kind, cnt, e_off=1)
    assert torch.equal(loss, loss2) and torch.equal(dU, dU2) and torch.equal(gp, gp2) and torch.equal(gn, gn2)


def test_bpr_triplet_fwd_bwd_equals_separate_kernels(ops):
    g = torch.Generator().manual_seed(12)
    n, D, RU, RI = 2048, 64, 300, 400
    Ut, It = torch.randn(RU, D, generator=g).cuda(), torch.randn(RI, D, generator=g).cuda()
    u = torch.randint(0, RU, (n,), generator=g).cuda()
    p, q = torch.randint(0, RI, (n,), generator=g).cuda(), torch.randint(0, RI, (n,), generator=g).cuda()
    loss, logits = ops.bpr_triplet_fwd(Ut, It, u, p, q)
    gu, gp, gn = ops.bpr_triplet_bwd(Ut, It, u, p, q, logits, None)
    loss2, gu2, gp2, gn2 = ops.bpr_triplet_fwd_bwd(Ut, It, u, p, q)
    assert torch.equal(loss, loss2) and torch.equal(gu, gu2) and torch.equal(gp, gp2) and torch.equal(gn, gn2)


@pytest.mark.parametrize("D", [64, 128, 10])
def test_sparse_adam_rows_matches_oracle(ops, D):
    """re_sparse_adam_rows: touched rows follow the oracle's sparse Adam (duplicates summed, padding row skipped, weight decay
    on touched rows), untouched rows keep parameters and moments bit for bit; two identical calls give identical bits."""
    from oracle import adam as oadam
    rng = np.random.default_rng(6)
    R, n = 5000, 3000
    W0 = rng.standard_normal((R, D)).astype(np.float32)
    m0 = (0.01 * rng.standard_normal((R, D))).astype(np.float32)
    v0 = (0.01 * rng.random((R, D))).astype(np.float32)
    idx = np.minimum(rng.zipf(1.3, n), R - 1).astype(np.int64)
    idx[::17] = 0                                   # padding row
    g = rng.standard_normal((n, D)).astype(np.float32)
    We, me, ve = W0.copy(), m0.copy(), v0.copy()
    oadam.sparse_adam_rows(We, me, ve, idx, g, 7, 1e-3, wd=1e-2, padding_idx=0)
    outs = []
    for _ in range(2):
        W, m, v = (torch.from_numpy(a.copy()).cuda() for a in (W0, m0, v0))
        ops.sparse_adam_rows(torch.from_numpy(g).cuda(), torch.from_numpy(idx).cuda(), W, m, v, 7, 1e-3, weight_decay=1e-2, padding_idx=0)
        outs.append((W.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    W, m, v = outs[0]
    np.testing.assert_allclose(W, We, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(m, me, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(v, ve, rtol=2e-5, atol=2e-6)
    untouched = np.setdiff1d(np.arange(R), idx[idx != 0])
    assert np.array_equal(W[untouched], W0[untouched]) and np.array_equal(m[untouched], m0[untouched]) and np.array_equal(v[untouched], v0[untouched])
