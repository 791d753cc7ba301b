"""GPU: SASRec on the engine vs the reference's golden vectors (tests/golden/sasrec_*.npz) and the oracle."""
import os

import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _engine(z, loss, **kw):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from aten_sasrec import AtenSASRec
    from recboard_amd.sasrec import SASRecEngine
    cls = AtenSASRec if kw.pop("encoder", "aten") == "aten" else SASRecEngine      # ("aten": the test comparator, tests/aten_sasrec.py)
    m = cls(int(z["cfg/N"]), 50, int(z["cfg/D"]), int(z["cfg/num_blocks"]), dropout_rate=0.0, loss=loss, **kw)
    m.load_state_dict({k[6:]: z[k] for k in z.files if k.startswith("param/") and z[k].dtype == np.float32})
    return m


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("loss", ["BCE", "BPR"])
def test_fit_loss_and_grads_match_reference_golden(loss):
    z = np.load(os.path.join(G, f"sasrec_{loss.lower()}.npz"))
    m = _engine(z, loss)
    seq, pos, neg = dev(z["in/seq"]), dev(z["in/pos"]), dev(z["in/neg"])
    A = m.arena
    for k, p in m.params.items():
        p.grad = A.view(A.grad, k)
    L = m.fit(seq, pos, neg)["rec_loss"]
    np.testing.assert_allclose(L.item(), float(z["out/rec_loss"]), rtol=1e-5)
    L.backward()
    for k, p in m.params.items():
        ref = z["grad/" + k]
        scale = max(np.abs(ref).max(), 1e-6)
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err <= 1e-4 * scale + 1e-7, (k, err, scale)   # north_star tolerance: 1e-4 relative
    assert (m.params["Item.embeddings.weight"].grad[0] == 0).all()   # padding row gets no gradient


@pytest.mark.parametrize("encoder,fixture", [("aten", "sasrec_bce.npz"), ("fused", "sasrec_bce.npz"), ("fused", "sasrec_bce_d128.npz")])
def test_encode_scores_topk_match_reference_golden(encoder, fixture):
    z = np.load(os.path.join(G, fixture))
    m = _engine(z, "BCE", encoder=encoder).eval()
    seq = dev(z["in/seq"])
    with torch.no_grad():
        u, _ = m.encode(seq)
    np.testing.assert_allclose(u.cpu().numpy(), z["out/userEmbds"], rtol=1e-4, atol=2e-5)
    sc = m.recommend_from_full(seq)
    np.testing.assert_allclose(sc.cpu().numpy(), z["out/scores"], rtol=1e-4, atol=2e-5)
    vals, idx = m.recommend_topk(seq, dev(z["in/seen_ptr"]), dev(z["in/seen_idx"]), 50)
    np.testing.assert_array_equal(idx.cpu().numpy(), z["out/topk_idx"])     # bit-exact top-K indices
    np.testing.assert_allclose(vals.cpu().numpy(), z["out/topk_vals"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("encoder,loss", [("aten", "BCE"), ("fused", "BCE"), ("fused", "BPR")])
def test_train_step_matches_oracle_adam_trajectory(encoder, loss):
    """3 full steps (zero_grad, backward, dense Adam with coupled L2) vs oracle fit + torch.optim.Adam on CPU."""
    from oracle import sasrec as osas
    z = np.load(os.path.join(G, f"sasrec_{loss.lower()}.npz"))
    m = _engine(z, loss, lr=5e-4, weight_decay=1e-6, encoder=encoder)
    P = osas.params_from_npz(z, requires_grad=True)
    opt = torch.optim.Adam(list(P.values()), lr=5e-4, betas=(0.9, 0.999), weight_decay=1e-6)
    seq, pos, neg = (torch.from_numpy(z[k]) for k in ("in/seq", "in/pos", "in/neg"))
    g0 = {}
    for step in range(3):
        l_gpu = m.train_step(seq.cuda(), pos.cuda(), neg.cuda())
        opt.zero_grad()
        l_cpu = osas.fit(P, seq, pos, neg, loss, 2)
        l_cpu.backward()
        for p in P.values():       # params absent from the graph still take the dense L2 + moment update
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        if step == 0:
            g0 = {k: p.grad.detach().abs().numpy().copy() for k, p in P.items()}
        opt.step()
        np.testing.assert_allclose(l_gpu.item(), l_cpu.item(), rtol=2e-5)
    # Adam's step is lr * m / (sqrt(v) + eps): a RELATIVE error of a gradient element becomes the same relative error of its step.
    # Gradients are held to 1e-4 of the tensor's largest element (the golden tests), i.e. an element of size |g| may be off by
    # 1e-4 gmax / |g| relatively -- 100 % for the elements that are zero in exact arithmetic (b_k: softmax is shift-invariant, what
    # any implementation computes there is its own rounding noise).  So every element is held to what that bound allows over the
    # three steps: 2e-5 + 1e-3 |theta| + 3 lr min(1, 2e-4 gmax / |g|).
    lr = 5e-4
    for k, p in m.params.items():
        a, b = p.detach().cpu().numpy(), P[k].detach().numpy()
        g = g0[k]
        slack = 3 * lr * np.minimum(1.0, 2e-4 * g.max() / np.maximum(g, 1e-30)) if g.max() > 0 else 0.0
        bad = np.abs(a - b) > 2e-5 + 1e-3 * np.abs(b) + slack
        assert not bad.any(), (k, int(bad.sum()), float(np.abs(a - b)[bad].max()))


@pytest.mark.parametrize("fixture", ["sasrec_bce.npz", "sasrec_bce_d128.npz"])
def test_fused_step_grads_match_reference_golden(fixture):
    """fused train step (no autograd): every gradient in the arena vs the reference's loss.backward() -- embedding_dim 64 (the
    benchmarked configuration) and 128 (BASELINE configs[4]; one row of the fixture is a full-length sequence: chained parts)."""
    z = np.load(os.path.join(G, fixture))
    m = _engine(z, "BCE", lr=0.0, encoder="fused")       # lr 0: parameters stay put, gradients stay in the arena
    seq, pos, neg = dev(z["in/seq"]), dev(z["in/pos"]), dev(z["in/neg"])
    L = m.train_step(seq, pos, neg)
    np.testing.assert_allclose(L.item(), float(z["out/rec_loss"]), rtol=1e-5)
    Gv = m.arena.views(m.arena.grad)
    for k in m.params:
        ref = z["grad/" + k]
        scale = max(np.abs(ref).max(), 1e-6)
        err = np.abs(Gv[k].cpu().numpy() - ref).max()
        assert err <= 1e-4 * scale + 1e-7, (k, err, scale)


def test_fused_step_with_dropout_trains():
    """dropout 0.5 (benchmark config): loss decreases over steps on a fixed batch, masks differ per step, nothing NaN."""
    z = np.load(os.path.join(G, "sasrec_bce.npz"))
    from recboard_amd.sasrec import SASRecEngine
    m = SASRecEngine(200, 50, 64, 2, dropout_rate=0.5, loss="BCE", lr=1e-3, weight_decay=1e-6, encoder="fused")
    seq, pos, neg = dev(z["in/seq"]), dev(z["in/pos"]), dev(z["in/neg"])
    losses = [m.train_step(seq, pos, neg).item() for _ in range(60)]
    assert all(np.isfinite(losses))
    assert np.mean(losses[-10:]) < np.mean(losses[:10]) - 0.05
    assert len({round(x, 6) for x in losses[:5]}) == 5


def test_fused_step_ce_loss_matches_reference_golden():
    """--loss CE (SASRec/main.py:216-219): logits over the whole catalog via the MFMA GEMM, row-wise CE in place."""
    z = np.load(os.path.join(G, "sasrec_ce.npz"))
    m = _engine(z, "CE", lr=0.0, encoder="fused")
    seq, pos, neg = dev(z["in/seq"]), dev(z["in/pos"]), dev(z["in/neg"])
    L = m.train_step(seq, pos, neg)
    np.testing.assert_allclose(L.item(), float(z["out/rec_loss"]), rtol=1e-5)
    Gv = m.arena.views(m.arena.grad)
    for k in m.params:
        ref = z["grad/" + k]
        scale = max(np.abs(ref).max(), 1e-6)
        err = np.abs(Gv[k].cpu().numpy() - ref).max()
        assert err <= 1e-4 * scale + 1e-7, (k, err, scale)


@pytest.mark.parametrize("loss", ["BCE", "BPR"])
def test_graph_replayed_step_is_identical_to_eager_step(loss):
    """train_step_graph (one staging launch + one hipGraph replay) must reproduce train_step_fused bit for bit, dropout
    included: the per-step seed and Adam's bias-correction scalars reach the captured kernels through device memory."""
    from recboard_amd.sasrec import SASRecEngine
    N, B, S = 500, 24, 50
    rng = np.random.default_rng(5)
    batches = []
    for _ in range(3):
        seq = rng.integers(1, N + 1, (B, S))
        for b in range(B):
            seq[b, : rng.integers(0, S - 1)] = 0
        pos, neg = rng.integers(0, N, (B, S)), rng.integers(0, N, (B, S))
        batches.append(tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg)))
    eager = SASRecEngine(N, S, 64, 2, dropout_rate=0.3, loss=loss, seed=3)
    graph = SASRecEngine(N, S, 64, 2, dropout_rate=0.3, loss=loss, seed=3)
    for i in range(6):
        b = batches[i % 3]
        le = eager.train_step_fused(*b).clone()
        lg = graph.train_step_graph(*b).clone()
        assert torch.equal(le, lg), (i, le, lg)
        assert torch.equal(eager.arena.data, graph.arena.data), i
    assert torch.equal(eager.arena.m, graph.arena.m) and torch.equal(eager.arena.v, graph.arena.v)


def test_graph_replayed_step_with_gradient_hook_matches_eager():
    """Data-parallel form: the captured graph ends before the optimizer, the hook (one all-reduce of the gradient arena in
    bench.py) runs between the replay and Adam.  A stand-in hook (halve the gradients) must give the eager result."""
    from recboard_amd.sasrec import SASRecEngine
    N, B, S = 300, 16, 50
    rng = np.random.default_rng(9)
    seq = rng.integers(1, N + 1, (B, S))
    for b in range(B):
        seq[b, : rng.integers(0, S - 1)] = 0
    batch = tuple(torch.from_numpy(a).cuda() for a in (seq, rng.integers(0, N, (B, S)), rng.integers(0, N, (B, S))))
    calls = []

    def hook(g):
        calls.append(1)
        g.mul_(0.5)

    eager = SASRecEngine(N, S, 64, 2, dropout_rate=0.2, loss="BCE", seed=4)
    graph = SASRecEngine(N, S, 64, 2, dropout_rate=0.2, loss="BCE", seed=4)
    for i in range(4):
        le = eager.train_step_fused(*batch, grad_hook=hook).clone()
        lg = graph.train_step_graph(*batch, grad_hook=hook).clone()
        assert torch.equal(le, lg), i
        assert torch.equal(eager.arena.data, graph.arena.data), i
    assert len(calls) == 8


@pytest.mark.parametrize("D", [64, 128])
def test_large_table_engine_first_step_equals_dense_engine(D):
    """SASRecLargeTableEngine (item table outside the arena, contribution rows -> row-sparse Adam) takes the same first step as
    the dense engine: from zero moments a row-sparse and a dense Adam agree on the touched rows, and with weight_decay = 0 the
    dense one leaves the untouched rows alone too."""
    from recboard_amd.large import SASRecLargeTableEngine
    from recboard_amd.sasrec import SASRecEngine
    N, B, S = 400, 12, 50
    rng = np.random.default_rng(21)
    seq = rng.integers(1, N + 1, (B, S))
    for b in range(B):
        seq[b, : rng.integers(0, S - 1)] = 0
    batch = tuple(torch.from_numpy(a).cuda() for a in (seq, rng.integers(0, N, (B, S)), rng.integers(0, N, (B, S))))
    dense = SASRecEngine(N, S, D, 2, dropout_rate=0.0, loss="BCE", lr=1e-2, seed=5, encoder="fused")   # (the same encoder kernels: Adam
    large = SASRecLargeTableEngine(N, S, D, 2, dropout_rate=0.0, loss="BCE", lr=1e-2, seed=5)           #  amplifies rounding differences)
    assert large.encoder == "fused"          # D = 64 and 128 both run the fused encoder kernels
    assert large.compact_rows and dense.compact_rows   # both run the compact-row item kernel (criterion inside): the same roundings
    assert large.split_long and dense.split_long       # ... on the same work items
    large.load_state_dict(dense.state_dict())
    ld = dense.train_step(*batch)
    ll = large.train_step(*batch)
    torch.testing.assert_close(ll, ld, rtol=1e-5, atol=1e-6)
    sd, sl = dense.state_dict(), large.state_dict()
    for k in sd:
        torch.testing.assert_close(sl[k], sd[k], rtol=2e-4, atol=2e-5, msg=k)
    # second step runs (moments of untouched rows now differ by design: dense Adam keeps decaying them)
    assert torch.isfinite(large.train_step(*batch))


def test_large_table_engine_graph_step_matches_eager_step():
    """The captured step of SASRecLargeTableEngine (torch block stack + engine kernels in one hipGraph, scalars through device
    words) reproduces the eager step (dropout 0: torch's RNG plays no part)."""
    from recboard_amd.large import SASRecLargeTableEngine
    N, B, S, D = 300, 8, 50, 128
    rng = np.random.default_rng(31)
    seq = rng.integers(1, N + 1, (B, S))
    for b in range(B):
        seq[b, : rng.integers(0, S - 1)] = 0
    batch = tuple(torch.from_numpy(a).cuda() for a in (seq, rng.integers(0, N, (B, S)), rng.integers(0, N, (B, S))))
    eager = SASRecLargeTableEngine(N, S, D, 2, dropout_rate=0.0, lr=1e-2, weight_decay=1e-4, seed=2)
    graph = SASRecLargeTableEngine(N, S, D, 2, dropout_rate=0.0, lr=1e-2, weight_decay=1e-4, seed=2)
    for i in range(3):
        le = eager.train_step(*batch).clone()
        lg = graph.train_step_graph(*batch).clone()
        torch.testing.assert_close(lg, le, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(graph.E, eager.E, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(graph.arena.data, eager.arena.data, rtol=1e-4, atol=1e-6)


def test_sharded_engine_on_one_rank_equals_large_table_engine():
    """SASRecShardedEngine (row-sharded item table, all-to-all lookups, gradient rows sent to their owners) with a process
    group of ONE rank over RCCL must take exactly the steps of SASRecLargeTableEngine on the same counter-initialised table;
    the two-rank exchange itself is covered under gloo (test_sharded_gloo.py)."""
    import socket
    import torch.distributed as dist
    from recboard_amd.large import SASRecLargeTableEngine, SASRecShardedEngine
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        N, B, S, D = 900, 10, 50, 128
        rng = np.random.default_rng(77)
        kw = dict(dropout_rate=0.0, lr=1e-2, weight_decay=1e-4, seed=6)
        # (a) the all-positions step (compact_rows = False: one contribution row per lookup, in lookup order) is the large-table engine's,
        #     bit for bit; (b) the default compact-row step sends the rows that exist, tagged with their lookups: same sums in another order
        large = SASRecLargeTableEngine(N, S, D, 2, table_init="counter", **kw)
        shard = SASRecShardedEngine(N, S, D, 2, dedup=False, capacity_factor=None, **kw)
        fx = SASRecShardedEngine(N, S, D, 2, capacity_factor=1.0, **kw)   # the sync-free form: owner bucketing on the device (re_route_bucket)
        for e in (large, shard, fx):
            e.compact_rows = False
        dd = SASRecShardedEngine(N, S, D, 2, capacity_factor=None, **kw)   # exact sizes: compact rows, distinct rows only, gradient rows pre-summed per sender
        cl = SASRecLargeTableEngine(N, S, D, 2, table_init="counter", **kw)                     # compact rows, unsharded
        cs = SASRecShardedEngine(N, S, D, 2, dedup=False, capacity_factor=None, **kw)            # compact rows, one row per lookup on the wire
        df = SASRecShardedEngine(N, S, D, 2, **kw)     # THE DEFAULT: fixed capacity 0.3 (sized for ~15 % real tokens; these batches are half real:
        ov = SASRecShardedEngine(N, S, D, 2, capacity_factor=0.02, **kw)   # every step overflows, is a no-op, and is re-run on the exact-size path)
        assert df.capacity_factor == 0.3
        cf = SASRecShardedEngine(N, S, D, 2, capacity_factor=1.0, **kw)                          # compact rows, fixed-capacity exchange
        cg = SASRecShardedEngine(N, S, D, 2, capacity_factor=1.0, **kw)                          # ... the same step as one hipGraph replay
        assert dd.compact_rows and cs.compact_rows and cs.split_long
        with pytest.raises(NotImplementedError, match="capacity_factor"):
            dd.train_step_graph(*(torch.zeros((B, S), dtype=torch.int64, device="cuda"),) * 3)
        assert torch.equal(large.E, shard.table.weight) and torch.equal(large.E, dd.table.weight)
        for step in range(3):
            seq = rng.integers(1, N + 1, (B, S))
            for b in range(B):
                seq[b, : rng.integers(0, S - 1)] = 0
            batch = tuple(torch.from_numpy(a).cuda() for a in (seq, rng.integers(0, N, (B, S)), rng.integers(0, N, (B, S))))
            ll = large.train_step(*batch)
            ls = shard.train_step(*batch)
            assert torch.equal(ll, ls), (step, ll, ls)
            assert torch.equal(large.arena.data, shard.arena.data), step
            assert torch.equal(large.E, shard.table.weight), step
            assert torch.equal(large.Em, shard.table.m) and torch.equal(large.Ev, shard.table.v), step
            lf = fx.train_step(*batch)
            fx.table.check_capacity()
            assert torch.equal(ll, lf) and torch.equal(large.E, fx.table.weight) and torch.equal(large.arena.data, fx.arena.data), step
            lc = cl.train_step(*batch)
            assert abs(float(lc) - float(ll)) <= 5e-5 * abs(float(ll)), step     # (compact rows: the one-tile-per-workgroup kernels' split arithmetic)
            for eng in (dd, cs, cf):
                le = eng.train_step(*batch)
                assert abs(float(le) - float(lc)) <= 1e-6 * abs(float(lc)), step
                # same sums in another association: Adam turns a last-bit difference of a near-zero gradient sum into a step of up to
                # lr, so a handful of entries may sit a few lr apart -- everything else agrees to rounding
                for a, b in ((eng.table.weight, cl.E), (eng.arena.data, cl.arena.data)):
                    diff = (a - b).abs()
                    assert float(diff.max()) <= 2.5e-2 * (step + 1), step
                    assert float((diff > 1e-5 * (1 + b.abs())).float().mean()) < 2e-3, step
            cf.table.check_capacity()
            cs.check_handover()
            # the captured step (lookup exchange, encoder step, gradient exchange, both optimizers in one graph) replays cf's launches
            lg = cg.train_step_graph(*batch)
            assert abs(float(lg) - float(lc)) <= 1e-6 * abs(float(lc)), step
            torch.testing.assert_close(cg.table.weight, cf.table.weight, rtol=1e-5, atol=1e-7)
            torch.testing.assert_close(cg.arena.data, cf.arena.data, rtol=1e-5, atol=1e-7)
            cg.table.check_capacity()
            df.train_step(*batch)
            ov.train_step_graph(*batch)
        # overflowing steps: nothing moved when the step ran; the re-runs (exact-size path, the skipped step's number and seed) give dd's table
        assert ov.overflow_steps == 1 and len(ov._pending) == 2 and ov.settle_overflow() == 3 and df.settle_overflow() == 3
        for eng in (df, ov):
            assert eng.arena.step == 3
            for a, b in ((eng.table.weight, dd.table.weight), (eng.arena.data, dd.arena.data)):
                diff = (a - b).abs()
                assert float(diff.max()) <= 7.5e-2 and float((diff > 1e-5 * (1 + b.abs())).float().mean()) < 2e-3
        seqs = torch.from_numpy(seq).cuda()
        sp = torch.arange(0, B + 1, device="cuda") * 3
        si = torch.sort(torch.from_numpy(rng.integers(0, N, (B, 3))).cuda(), 1).values.reshape(-1)
        v1, i1 = large.recommend_topk(seqs, sp, si, 20)
        v2, i2 = shard.recommend_topk(seqs, sp, si, 20)
        assert torch.equal(i1, i2) and torch.equal(v1, v2)
    finally:
        for g_ in ("cg", "ov"):
            if g_ in locals():
                locals()[g_].release_graphs()        # (a live graph holding captured RCCL work blocks the group's teardown)
        dist.destroy_process_group()


def test_ce_in_catalog_chunks_equals_materialised_ce():
    """loss='CE' with the catalog walked in column chunks (re_ce_chunk_*: online log-sum-exp, logits recomputed in the backward)
    against the single-chunk form that materialises [M, N]: same loss and same parameters after two steps."""
    from recboard_amd.sasrec import SASRecEngine
    N, B, S = 700, 24, 50
    rng = np.random.default_rng(3)
    seq = rng.integers(1, N + 1, (B, S))
    for b in range(B):
        seq[b, : rng.integers(0, S - 1)] = 0
    pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    batch = tuple(torch.from_numpy(a).cuda() for a in (seq, pos, np.zeros_like(pos)))
    eng = []
    for limit in (1 << 28, 4 * 600 * 96):          # one chunk; chunks of ~96 columns (the last one ragged)
        m = SASRecEngine(N, S, 64, 2, dropout_rate=0.0, loss="CE", lr=1e-3, seed=2)
        m.ce_logits_bytes = limit
        eng.append((m, float(m.train_step(*batch))))
    assert abs(eng[0][1] - eng[1][1]) <= 2e-6 * abs(eng[0][1])
    # the gradients of the step (compared before Adam: its first step turns the rounding noise of exactly-zero gradients -- the key
    # bias -- into +-lr)
    ga, gb = eng[0][0].arena.views(eng[0][0].arena.grad), eng[1][0].arena.views(eng[1][0].arena.grad)
    for k in ga:
        torch.testing.assert_close(gb[k], ga[k], rtol=1e-4, atol=1e-7, msg=k)


@pytest.mark.parametrize("in_tail", [True, False])
def test_pipelined_preparation_gives_the_same_steps(in_tail):
    """train_step_graph(next_batch=...): the next batch prepared during the step -- by jobs of the step's tail launch (in_tail) or by a
    preparation launch on a side stream --, two captured copies alternating: the same losses and parameters as the plain captured step
    (dropout on: the per-step seeds must line up too)."""
    from recboard_amd.sasrec import SASRecEngine
    N, B, S = 500, 48, 50
    rng = np.random.default_rng(12)
    batches = []
    for _ in range(5):
        seq = np.zeros((B, S), np.int64)
        for b in range(B):
            n = int(rng.integers(1, S))
            seq[b, S - n:] = rng.integers(1, N + 1, n)
        pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
        neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
        batches.append(tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg)))
    a = SASRecEngine(N, S, 64, 2, dropout_rate=0.3, lr=1e-3, seed=4)
    b = SASRecEngine(N, S, 64, 2, dropout_rate=0.3, lr=1e-3, seed=4)
    a.prep_in_tail = False
    b.prep_in_tail = in_tail
    for i in range(9):
        la = a.train_step_graph(*batches[i % 5]).clone()
        # (step 4 is called WITHOUT the batch the step before announced, step 5 announces nothing: both must fall back to a preparation in front)
        nxt = batches[(i + 1) % 5] if i not in (5, 8) else None
        cur = batches[i % 5] if i != 4 else tuple(t.clone() for t in batches[i % 5])
        lb = b.train_step_graph(*cur, next_batch=nxt).clone()
        if in_tail:
            assert torch.equal(lb, la), i
        else:
            torch.testing.assert_close(lb, la, rtol=1e-5, atol=1e-7)
    if in_tail:
        assert torch.equal(b.arena.data, a.arena.data)
        assert len(b._tail_pipes) == 1 and int(b._ticket.abs().sum().item()) == 0
    else:
        assert b._staged is None and len([k for k in b._graphs if len(k) == 5]) == 2
        torch.testing.assert_close(b.arena.data, a.arena.data, rtol=1e-4, atol=1e-6)
    b.check_handover()


@pytest.mark.parametrize("B", [1536, 4096])
def test_tail_preparation_at_large_batches_gives_the_same_steps(B):
    """From 1 024 sequences on the step's tail launch shares the next plan's spans phase among its last workgroups (csrc/enc_plan_body.h:
    PL_MODE_SPANS with nparts > 1 -- arrival and token counters, the last to arrive publishes), cuts the weight-gradient contractions 40-way
    and deals the position jobs by ranges of sequences: the same losses and parameters, bit for bit, as the plain captured step whose batch a
    preparation launch in front prepares -- over several steps, so that the counters' return to zero is exercised."""
    import bench
    from recboard_amd.sasrec import SASRecEngine
    cfg = dict(bench.BEAUTY, B=B)
    bs = [tuple(torch.from_numpy(x).cuda() for x in b) for b in bench.synth_batches(cfg, 4, 3)]
    a = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=2)
    b = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=2)
    a.prep_in_tail = False
    b.prep_in_tail = True
    for i in range(7):
        la = a.train_step_graph(*bs[i % 4]).clone()
        lb = b.train_step_graph(*bs[i % 4], next_batch=bs[(i + 1) % 4]).clone()
        assert torch.equal(lb, la), i
    assert torch.equal(b.arena.data, a.arena.data)
    b.check_handover()
    assert int(b._ticket.abs().sum().item()) == 0


@pytest.mark.parametrize("pipelined", [False, True])
def test_captured_step_is_reproducible_from_process_to_process(pipelined):
    """Two fresh processes, the same seeds and Beauty-shaped batches, 160 captured steps each: the same parameters bit for bit.  (Rounds 3 - 5
    ran the tile kernels at one workgroup per CU for this test's sake: with two, about every second pair of runs parted in the last digits.
    Round 6 traced that to packed-fp32 instructions with two waves on a SIMD -- enc_tile.hip is built without them, two per CU is the
    product's setting at D = 64: profiles/r6_handover_notes.txt, and the in-process form of this test below.)"""
    import re
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(G), "..", "scripts", "determinism.py")
    out = []
    for _ in range(2):
        env = dict(os.environ, **({"DET_NEXT": "1"} if pipelined else {}))          # (pipelined: the next batch prepared by the tail launch)
        r = subprocess.run([sys.executable, script, "160", "64"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "parameters identical: True" in r.stdout, r.stdout
        out.append(re.search(r"sha1 of the parameters: (\w+)", r.stdout).group(1))
    assert out[0] == out[1], out


@pytest.mark.parametrize("B", [2048, 4096])
def test_two_tile_workgroups_per_cu_replay_the_same_bits(B):
    """The large-batch regime (B >= 1 024: the looped tile kernels, two workgroups resident per CU since round 6): the SAME captured step
    replayed twelve times -- lr = 0, one batch, one dropout seed -- gives the same loss and gradient arena every time.  With packed-fp32
    instructions in the tile kernels this failed in five replays of six (one lane group of one register of the backward's incoming
    gradient; profiles/r6_handover_notes.txt), so the test also pins what the library was built as."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import hashlib
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    from recboard_amd.lib import load as lib_load
    from recboard_amd.sasrec import SASRecEngine
    assert lib_load().re_tile_wgs_per_cu(64) == 2 and lib_load().re_tile_wgs_per_cu(128) == 1
    cfg = dict(bench.BEAUTY, B=B)
    seq, pos, neg = (torch.from_numpy(x).cuda() for x in bench.synth_batches(cfg, 1, 1)[0])
    m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=0.0, weight_decay=0.0, seed=1)
    seen = set()
    for _ in range(12):
        m.arena.step = 0                                     # (the same dropout seed every time)
        loss = m.train_step_graph(seq, pos, neg)
        seen.add(hashlib.sha1(loss.cpu().numpy().tobytes() + m.arena.grad.cpu().numpy().tobytes()).hexdigest())
    m.check_handover()
    assert len(seen) == 1, f"{len(seen)} different results in 12 replays"


@pytest.mark.parametrize("captured", [True, False])
def test_one_launch_tail_equals_the_two_branch_form(captured):
    """re_sasrec_step_tail (the scatter-add's workgroups take the weight-gradient jobs) against re_scatter_adam_rows_small beside
    re_sasrec_encoder_step_part(part = 4) on two streams: the same parameters and losses, bit for bit (dropout on, Adam fused or not)."""
    from recboard_amd.sasrec import SASRecEngine
    N, B, S = 700, 96, 50
    rng = np.random.default_rng(21)
    batches = []
    for _ in range(3):
        seq = np.zeros((B, S), np.int64)
        for b in range(B):
            n = int(rng.integers(1, S + 1))
            seq[b, S - n:] = rng.integers(1, N + 1, n)
        pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
        neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
        batches.append(tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg)))
    eng = []
    for tail in (True, False):
        m = SASRecEngine(N, S, 64, 2, dropout_rate=0.3, lr=1e-3, weight_decay=1e-6, seed=9)
        m.fuse_tail = tail
        losses = [(m.train_step_graph(*batches[i % 3]) if captured else m.train_step(*batches[i % 3], None)).clone() for i in range(5)]
        eng.append((m, torch.stack(losses)))
    assert torch.equal(eng[0][1], eng[1][1])
    assert torch.equal(eng[0][0].arena.data, eng[1][0].arena.data)
    assert torch.equal(eng[0][0].arena.grad, eng[1][0].arena.grad)
    assert int(eng[0][0]._ticket.abs().sum().item()) == 0


@pytest.mark.parametrize("D", [64, 128])
def test_large_table_one_launch_tail_equals_the_two_branch_form(D):
    """re_sasrec_step_tail_sparse (the row-sparse Adam's workgroups take the weight-gradient jobs) against re_sparse_adam_rows_small beside
    re_sasrec_encoder_step_part(part = 4) on two streams: the same table, moments, parameters and losses, bit for bit."""
    from recboard_amd.large import SASRecLargeTableEngine
    N, B, S = 5000, 64, 50
    rng = np.random.default_rng(33)
    batches = []
    for _ in range(3):
        seq = np.zeros((B, S), np.int64)
        for b in range(B):
            n = int(rng.integers(1, S + 1))
            seq[b, S - n:] = np.minimum(rng.zipf(1.3, n), N)          # (a heavy head: the whole-workgroup path of the row-sparse Adam)
        pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
        neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
        batches.append(tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg)))
    eng = []
    for tail in (True, False):
        m = SASRecLargeTableEngine(N, S, D, 2, dropout_rate=0.3, lr=1e-3, weight_decay=1e-6, seed=4)
        m.fuse_tail = tail
        losses = [m.train_step_graph(*batches[i % 3]).clone() for i in range(5)]
        eng.append((m, torch.stack(losses)))
    assert torch.equal(eng[0][1], eng[1][1])
    for name in ("E", "Em", "Ev"):
        assert torch.equal(getattr(eng[0][0], name), getattr(eng[1][0], name)), name
    assert torch.equal(eng[0][0].arena.data, eng[1][0].arena.data)
    assert int(eng[0][0]._ticket.abs().sum().item()) == 0


def test_large_table_pipelined_preparation_gives_the_same_steps():
    """SASRecLargeTableEngine.train_step_graph(next_batch=...): the next batch prepared by jobs of the step's tail launch
    (re_sasrec_step_tail_sparse + re_next_prep) -- the same table, parameters and losses as the plain captured step, bit for bit."""
    from recboard_amd.large import SASRecLargeTableEngine
    N, B, S, D = 4000, 64, 50, 128
    rng = np.random.default_rng(35)
    batches = []
    for _ in range(4):
        seq = np.zeros((B, S), np.int64)
        for b in range(B):
            n = int(rng.integers(1, S + 1))
            seq[b, S - n:] = np.minimum(rng.zipf(1.3, n), N)
        pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
        neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
        batches.append(tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg)))
    a = SASRecLargeTableEngine(N, S, D, 2, dropout_rate=0.3, lr=1e-3, weight_decay=1e-6, seed=4)
    b = SASRecLargeTableEngine(N, S, D, 2, dropout_rate=0.3, lr=1e-3, weight_decay=1e-6, seed=4)
    for i in range(7):
        la = a.train_step_graph(*batches[i % 4]).clone()
        lb = b.train_step_graph(*batches[i % 4], next_batch=batches[(i + 1) % 4] if i != 3 else None).clone()
        assert torch.equal(lb, la), i
    for name in ("E", "Em", "Ev"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(a.arena.data, b.arena.data)
    assert len(b._tail_pipes) == 1 and not getattr(a, "_tail_pipes", {})


def test_pipelined_step_with_a_gradient_hook_matches_the_plain_one():
    """The data-parallel form (gradient hook, then Adam, behind the replay) through the tail-prepared pipeline: the same losses and parameters
    as the same steps with every batch prepared in front of its step."""
    from recboard_amd.sasrec import SASRecEngine
    N, B, S = 500, 48, 50
    rng = np.random.default_rng(13)
    batches = []
    for _ in range(4):
        seq = np.zeros((B, S), np.int64)
        for b in range(B):
            n = int(rng.integers(1, S))
            seq[b, S - n:] = rng.integers(1, N + 1, n)
        pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
        neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
        batches.append(tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg)))
    calls = []
    hook = lambda g: (calls.append(1), g.mul_(0.5))[0]          # (a stand-in for the all-reduce: something that changes the gradients)
    a = SASRecEngine(N, S, 64, 2, dropout_rate=0.3, lr=1e-3, seed=4)
    b = SASRecEngine(N, S, 64, 2, dropout_rate=0.3, lr=1e-3, seed=4)
    for i in range(6):
        la = a.train_step_graph(*batches[i % 4], grad_hook=hook).clone()
        lb = b.train_step_graph(*batches[i % 4], grad_hook=hook, next_batch=batches[(i + 1) % 4]).clone()
        assert torch.equal(la, lb), i
    assert torch.equal(a.arena.data, b.arena.data) and len(calls) == 12
    assert any(len(k) == 4 for k in b._tail_pipes)
