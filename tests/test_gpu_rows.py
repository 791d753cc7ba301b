"""GPU: the fused SASRec step's compact-row pieces through the C ABI -- the loss head on the plan's rows (re_sasrec_loss_rows)
and the sort-free dense table gradient (re_scatter_add_rows_small) -- against the all-positions kernels / torch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(B, S, N, seed, full=False):
    rng = np.random.default_rng(seed)
    lens = np.full(B, S) if full else np.clip(rng.geometric(1 / 5.9, B) + 1, 1, S)
    seq = np.zeros((B, S), np.int64)
    for b in range(B):
        seq[b, S - lens[b]:] = rng.integers(1, N + 1, lens[b])
    pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    return tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg))


@pytest.mark.parametrize("D,R,n,hot", [(64, 12102, 13000, False), (64, 50, 9000, True), (128, 3000, 4100, False), (64, 12102, 0, False),
                                       (64, 90000, 20000, False)])
def test_scatter_small_matches_index_add(D, R, n, hot):
    from recboard_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    stride = (max(n, 16) + 15) // 16 * 16 + 32
    keys = torch.zeros((3, stride), dtype=torch.int32, device="cuda")
    if n:
        k = torch.randint(0, R, (3, n), device="cuda", generator=g, dtype=torch.int32)
        if hot:
            k[:, ::3] = 7                                     # one row takes a third of everything
        k[0, :5] = 0                                          # padding entries
        keys[:, :n] = k
        keys[:, n:] = 5                                       # beyond n: must not be read as contributions
    G = torch.randn(3, stride, D, device="cuda", generator=g)
    out = torch.full((R, D), 7.0, device="cuda")
    ops.scatter_add_rows_small(G, keys, R, out, n_regions=3, region_stride=stride, n=n, padding_idx=0, scale=0.5)
    ref = torch.zeros(R, D, device="cuda", dtype=torch.float64)
    if n:
        kk = keys[:, :n].reshape(-1).long()
        rows = G[:, :n].reshape(-1, D).double() * 0.5
        keep = kk != 0
        ref.index_add_(0, kk[keep], rows[keep])
    # (fp32 sums of up to 9 000 terms in the hot case: the bound scales with the number of terms)
    assert torch.allclose(out.double(), ref, rtol=1e-5, atol=2e-3 if hot else 1e-5)
    assert float(out[0].abs().max()) == 0.0
    # n from device memory (tiles * 16), and bitwise reproducible
    if n and n % 16 == 0:
        nd = torch.tensor([n // 16], dtype=torch.int32, device="cuda")
        out2 = torch.empty_like(out)
        ops.scatter_add_rows_small(G, keys, R, out2, n_regions=3, region_stride=stride, n_dev=nd, n_mul=16, padding_idx=0, scale=0.5)
        assert torch.equal(out, out2)
    out3 = torch.empty_like(out)
    ops.scatter_add_rows_small(G, keys, R, out3, n_regions=3, region_stride=stride, n=n, padding_idx=0, scale=0.5)
    assert torch.equal(out, out3)


def test_scatter_small_rejects_what_it_cannot_do():
    from recboard_amd import ops
    with pytest.raises(RuntimeError):
        ops.scatter_add_rows_small(torch.zeros(16, 32, device="cuda"), torch.zeros(16, dtype=torch.int32, device="cuda"), 10,
                                   torch.zeros(10, 32, device="cuda"))
    with pytest.raises(RuntimeError):      # past ~100 k rows the sorted path is the one to use
        ops.scatter_add_rows_small(torch.zeros(16, 64, device="cuda"), torch.zeros(16, dtype=torch.int32, device="cuda"), 1 << 20,
                                   torch.zeros(1 << 20, 64, device="cuda"))


@pytest.mark.parametrize("D,kind,full", [(64, 0, False), (64, 1, False), (128, 0, False), (64, 0, True)])
def test_loss_rows_matches_all_positions_kernel(D, kind, full):
    from recboard_amd import ops
    B, S, N = 96, 50, 500
    seq, pos, neg = _batch(B, S, N, 5, full)
    seq[3, 40] = 0                                            # a pad INSIDE a sequence: a row exists, it is not a valid position
    pos[3, 40] = 0; neg[3, 40] = 0
    g = torch.Generator(device="cuda").manual_seed(1)
    U = torch.randn(B * S, D, device="cuda", generator=g)
    E = torch.randn(N + 1, D, device="cuda", generator=g) * 0.3
    pb = ops.sasrec_batch_prep(seq, pos, neg, max_tiles=ops.max_tiles(D))
    NR = ops.sasrec_plan_rows(B, S)
    dUr = torch.full((NR, D), 9.0, device="cuda")
    Gr = torch.full((3, NR, D), 9.0, device="cuda")
    keys = torch.full((3, NR), -7, dtype=torch.int32, device="cuda")
    loss = ops.sasrec_loss_rows(U, E, pb.seq, pb.pos, pb.neg, pb.plan, kind, pb.count, dUr, Gr, keys, e_off=1)
    loss2 = ops.sasrec_loss_rows(U, E, pb.seq, pb.pos, pb.neg, pb.plan, kind, pb.count, dUr, Gr, keys, e_off=1)
    assert torch.equal(loss, loss2)                           # last-workgroup reduction: fixed order
    n = B * S
    ref_loss, dU, gp, gn = ops.pair_loss_fwd_bwd(U, E, pos.reshape(-1), neg.reshape(-1), pb.valid, kind, pb.count, e_off=1)
    assert abs(float(loss) - float(ref_loss)) <= 2e-6 * abs(float(ref_loss))
    hdr = pb.plan.view(torch.int32)[:8].cpu().numpy()
    nr = 16 * int(hdr[1])
    off = int(lib_rowmap_word(B, S))
    rm = pb.plan.view(torch.int32)[off: off + 2 * nr].view(nr, 2).cpu().numpy()
    gid = torch.from_numpy(rm[:, 0].astype(np.int64)).cuda()
    live = gid >= 0
    gl = gid[live]
    seqf, posf, negf = seq.reshape(-1), pos.reshape(-1), neg.reshape(-1)
    # every valid position has exactly one row
    assert int((seqf != 0).sum()) == int((seqf[gl] != 0).sum())
    assert torch.equal(dUr[:nr][live], dU[gl])
    valid = seqf[gl] != 0
    assert torch.equal(Gr[1, :nr][live][valid], gp[gl][valid]) and torch.equal(Gr[2, :nr][live][valid], gn[gl][valid])
    k = keys[:, :nr]
    assert torch.equal(k[0][live].long(), seqf[gl])
    assert torch.equal(k[1][live].long(), torch.where(valid, posf[gl] + 1, torch.zeros_like(gl)))
    assert torch.equal(k[2][live].long(), torch.where(valid, negf[gl] + 1, torch.zeros_like(gl)))
    assert int(k[:, ~live.cpu().numpy()].abs().sum()) == 0 if (~live).any() else True


def lib_rowmap_word(B, S):
    mt = B * ((S + 15) // 16)
    return (8 + mt + 1) // 2 * 2


@pytest.mark.parametrize("D,kind,p", [(64, 0, 0.0), (64, 1, 0.5), (128, 0, 0.2)])
def test_forward_with_loss_head_matches_forward_then_loss_rows(D, kind, p):
    """re_sasrec_encoder_fwd_loss == re_sasrec_encoder_fwd + re_sasrec_loss_rows (u and the tape bitwise; the criterion's dots are
    reduced in another order inside the item: 1e-6)."""
    from recboard_amd import ops
    from recboard_amd.sasrec import SASRecEngine
    B, S, N, L = 64, 50, 300, 2
    m = SASRecEngine(N, S, D, L, dropout_rate=p, loss="BCE" if kind == 0 else "BPR", seed=3)
    seq, pos, neg = _batch(B, S, N, 9)
    pb = m.prepare_batch(seq, pos, neg)
    P = m.params
    E, Pp = P["Item.embeddings.weight"].detach(), P["Position.weight"].detach()
    lw, lb = P["lastLN.weight"].detach(), P["lastLN.bias"].detach()
    bt = m._block_tensors()
    NR = ops.sasrec_plan_rows(B, S)
    Lb = __import__("recboard_amd.lib", fromlist=["load"]).load()
    tb = Lb.re_sasrec_tape_bytes(B, S, D, L) // 4
    outs = []
    for fused in (False, True):
        u = torch.zeros(B, S, D, device="cuda")
        tape = torch.zeros(tb, device="cuda")
        dUr = torch.zeros(NR, D, device="cuda"); Gr = torch.zeros(3, NR, D, device="cuda")
        keys = torch.zeros(3, NR, dtype=torch.int32, device="cuda")
        ws = torch.zeros(256, dtype=torch.uint8, device="cuda")
        if fused:
            loss = ops.sasrec_encoder_fwd_loss(E, Pp, pb.seq, pb.pos, pb.neg, float(D ** 0.5), bt, lw, lb, L, p, 77, pb.plan, kind, pb.count,
                                               u, tape, dUr, Gr, keys, ws)
            loss_b = ops.sasrec_encoder_fwd_loss(E, Pp, pb.seq, pb.pos, pb.neg, float(D ** 0.5), bt, lw, lb, L, p, 77, pb.plan, kind,
                                                 pb.count, u, tape, dUr, Gr, keys, ws)
            assert torch.equal(loss, loss_b) and int(ws.view(torch.int64)[0]) == 0
        else:
            ops.sasrec_embed_encoder_fwd(E, Pp, pb.seq, float(D ** 0.5), bt, lw, lb, L, p, 77, need_tape=True, out=u, tape=tape, plan=pb.plan)
            loss = ops.sasrec_loss_rows(u.view(-1, D), E, pb.seq, pb.pos, pb.neg, pb.plan, kind, pb.count, dUr, Gr, keys, e_off=1)
        outs.append((u, tape, float(loss), dUr, Gr, keys))
    a, b = outs
    nr = 16 * int(pb.plan.view(torch.int32)[1])
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert abs(a[2] - b[2]) <= 2e-6 * abs(a[2])
    assert torch.equal(a[5][:, :nr], b[5][:, :nr])
    assert torch.allclose(a[3][:nr], b[3][:nr], rtol=2e-5, atol=1e-9)
    live = (a[5][1, :nr] != 0)
    assert torch.allclose(a[4][1:, :nr][:, live], b[4][1:, :nr][:, live], rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize("D,p", [(128, 0.2)])
def test_one_launch_item_kernel_equals_two_launches(D, p):
    """re_sasrec_encoder_step (forward + criterion + backward per work item in one launch) against re_sasrec_encoder_fwd_loss +
    re_sasrec_encoder_bwd: the same parameters after three Adam steps, bit for bit.  (The workgroup-per-item kernels, `tile_step` off:
    with it on the one-launch step is the one-tile-per-workgroup kernel, whose arithmetic differs -- the tests below.)"""
    from recboard_amd.sasrec import SASRecEngine
    B, S, N = 96, 50, 700
    eng = []
    for fused in (True, False):
        m = SASRecEngine(N, S, D, 2, dropout_rate=p, loss="BCE", lr=1e-3, weight_decay=1e-6, seed=4)
        m.fused_item_kernel = fused
        m.tile_step = False
        losses = []
        for i in range(3):
            seq, pos, neg = _batch(B, S, N, 30 + i, full=(i == 2))
            losses.append(float(m.train_step(seq, pos, neg)))
        eng.append((m, losses))
    assert eng[0][1] == eng[1][1]
    assert torch.equal(eng[0][0].arena.data, eng[1][0].arena.data)


@pytest.mark.parametrize("loss,p,ncu,D", [("BCE", 0.5, None, 64), ("BPR", 0.0, None, 64), ("BCE", 0.3, 24, 64), ("BCE", 0.3, None, 128)])
def test_wave_per_tile_step_matches_oracle(loss, p, ncu, D):
    """D = 64: re_sasrec_encoder_step runs four waves per tile (csrc/enc_tile.hip: activations in registers, bf16 hi / mid split products
    on the XDL pipe) -- against the CPU oracle with the same dropout masks: loss to 2e-5, every gradient to the 1e-4 bound; short
    sequences sharing tiles, sequences of 2 - 4 tiles (k, v and the partial dK, dV cross waves), a full-length batch, and (ncu = 24)
    plans whose items hold several tiles.  The oracle takes the engine's relu gates where its own pre-activation is within 2e-5 of
    zero (oracle/sasrec.py: block).  Twice the same bits."""
    from oracle import sasrec as osas
    from recboard_amd import ops
    from recboard_amd.sasrec import SASRecEngine
    B, S, N, L = 96, 50, 700, 2
    for i in range(3):
        seq, pos, neg = _batch(B, S, N, 30 + i, full=(i == 2))
        if i == 1:                       # a few sequences of every tile count next to the short ones
            for b, n in enumerate((49, 40, 33, 20, 17)):
                seq[b] = 0
                seq[b, S - n:] = torch.arange(1, n + 1, device=seq.device)
                pos[b] = torch.where(seq[b] > 0, (seq[b] * 7) % N, 0)
                neg[b] = torch.where(seq[b] > 0, (seq[b] * 13 + 5) % N, 0)
        runs = []
        for rep in range(2):
            m = SASRecEngine(N, S, D, L, dropout_rate=p, loss=loss, lr=0.0, weight_decay=0.0, seed=4 + i)
            if ncu:
                m._plan_ncu = lambda: ncu
            assert m._wave_step()
            with torch.no_grad():            # non-trivial biases / LayerNorm parameters
                g = torch.Generator().manual_seed(3)
                for k, q in m.params.items():
                    if k.endswith("bias"):
                        q.copy_((0.05 * torch.randn(q.shape, generator=g)).cuda())
                    elif "LN" in k:
                        q.copy_((1.0 + 0.1 * torch.randn(q.shape, generator=g)).cuda())
            sd = {k: v.cpu() for k, v in m.state_dict().items()}
            pb = m.prepare_batch(seq, pos, neg)
            if ncu and i < 2:
                w = pb.plan.view(torch.int32)
                assert int(w[3]) > 1         # several tiles per short item (the full-length batch has no short items)
            seed = m._step_seed()
            lval = float(m.train_step(seq, pos, neg, aux=pb))
            runs.append((lval, m.arena.grad.clone()))
        assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
        tape = m._buffers(B, S)["tape"]
        gates = {l: ((ops.sasrec_tape_array(tape, pb.plan, B, S, D, L, "HR", l) > 0).cpu(), 2e-5) for l in range(L)}
        P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ref = osas.fit(P, seq.cpu(), pos.cpu(), neg.cpu(), loss, L, drop=dict(p=p, seed=seed) if p > 0 else None, gates=gates)
        ref.backward()
        assert abs(lval - ref.item()) <= 2e-5 * abs(ref.item())
        Gv = m.arena.views(m.arena.grad)
        for k, q in P.items():
            r = q.grad if q.grad is not None else torch.zeros_like(q)
            err = (Gv[k].cpu() - r).abs().max().item()
            assert err <= 1e-4 * r.abs().max().item() + 1e-7, (i, k, err, r.abs().max().item())


@pytest.mark.parametrize("D", [64, 128])
def test_long_sequences_split_over_two_workgroups_match_whole_items(D):
    """split_long: a sequence of 3 - 4 tiles runs as two work items in two workgroups that hand k, v and the partial dK, dV over
    through the tape under flags.  Same function as the whole-item plan: loss and every gradient within rounding (the attention
    products see their key tiles in another grouping), twice the same bits, and the flags come back clean."""
    from recboard_amd import ops
    from recboard_amd.sasrec import SASRecEngine
    B, S, N = 48, 50, 400
    rng = np.random.default_rng(12)
    lens = np.concatenate([rng.integers(33, 50, 16), rng.integers(17, 33, 8), np.clip(rng.geometric(1 / 5.9, B - 24) + 1, 1, 16)])
    seq = np.zeros((B, S), np.int64)
    for b in range(B):
        seq[b, S - lens[b]:] = rng.integers(1, N + 1, lens[b])
    pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    batch = tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg))
    out = []
    for split in (True, False, True):
        m = SASRecEngine(N, S, D, 2, dropout_rate=0.3, loss="BCE", lr=1e-3, seed=8)
        m.split_long = split
        m.tile_step = False                  # (the workgroup-per-item kernels: the one-tile-per-workgroup step never splits)
        if D == 64:
            m.fused_item_kernel = False
        pb = m.prepare_batch(*batch)
        kinds = (pb.plan.view(torch.int32)[8:8 + int(pb.plan.view(torch.int32)[0])].cpu().numpy() >> 28) & 15
        assert (set(kinds.tolist()) >= {2, 3}) == split, kinds
        loss = float(m.train_step(*batch, aux=pb))
        W = m._buffers(B, S)
        flags = W["tape"][-(B * 4 * 8 + 16):].view(torch.int32)
        assert int(flags[:-15].abs().sum()) == 0    # every flag consumed and cleared, no time-out (the word behind the error word is the tile kernels' launch epoch)
        out.append((loss, m.arena.grad.clone()))
    assert out[0][0] == out[2][0] and torch.equal(out[0][1], out[2][1])
    assert abs(out[0][0] - out[1][0]) <= 2e-6 * abs(out[1][0])
    torch.testing.assert_close(out[0][1], out[1][1], rtol=2e-4, atol=1e-7)


def test_split_is_refused_when_the_halves_could_not_all_be_resident():
    """More long sequences than compute units: the two halves of a split sequence must be resident at once, so the plan keeps the
    sequences whole (and the step still equals the split_long = False step bit for bit)."""
    from recboard_amd.sasrec import SASRecEngine
    B, S, N = 320, 50, 300
    rng = np.random.default_rng(4)
    seq = rng.integers(1, N + 1, (B, S))
    seq[:, : S - 40] = 0                                   # every sequence has 40 rows = 3 tiles
    pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    batch = tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg))
    out = []
    for split in (True, False):
        m = SASRecEngine(N, S, 64, 2, dropout_rate=0.2, loss="BPR", lr=1e-3, seed=3)
        m.split_long = split
        pb = m.prepare_batch(*batch)
        w = pb.plan.view(torch.int32)
        kinds = (w[8:8 + int(w[0])].cpu().numpy() >> 28) & 15
        assert set(kinds.tolist()) == {1}
        out.append((float(m.train_step(*batch, aux=pb)), m.arena.grad.clone()))
        m.check_handover()
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])



@pytest.mark.parametrize("D,case", [(64, "zipf"), (128, "zipf"), (128, "hot"), (64, "one_row"), (128, "regions_dev"), (64, "empty")])
def test_sparse_adam_rows_small_matches_oracle(D, case):
    """re_sparse_adam_rows_small (csrc/adam_rows.hip: owner-computes, no sort) against the oracle's sparse Adam: duplicates summed, padding /
    out-of-range keys dropped, untouched rows bit-identical, two calls identical; `one_row` (every key the same row: more than an LDS
    list holds) takes the kernel's slow exact path; `regions_dev`: int32 keys in three regions, the live length from device memory."""
    from oracle import adam as oadam
    from recboard_amd import ops
    rng = np.random.default_rng(11)
    R = 300_000
    W0 = rng.standard_normal((R, D)).astype(np.float32)
    m0 = (0.01 * rng.standard_normal((R, D))).astype(np.float32)
    v0 = (0.01 * rng.random((R, D))).astype(np.float32)
    kw = {}
    if case == "zipf":
        n = 14000
        keys = np.minimum(rng.zipf(1.05, n), R - 1).astype(np.int64)
        keys[::17] = 0                                  # padding row
        keys[5], keys[9] = -3, R + 4                    # out of range
    elif case == "hot":
        n = 9000
        keys = rng.integers(1, R, n).astype(np.int64)
        keys[rng.random(n) < 0.3] = 777                 # ~2 700 contributions to one row: a long run, summed by the whole workgroup
        keys[rng.random(n) < 0.01] = 4242               # ~90: a long run next to it
    elif case == "one_row":
        n = 5000
        keys = np.full(n, 123456, np.int64)
    elif case == "empty":
        n = 100
        keys = np.zeros(n, np.int64)
    else:
        stride, live = 4096, 16 * 130                   # three regions of 4096 entries, the first 2080 of each live
        keys = np.zeros((3, stride), np.int32)
        keys[:, :live] = np.minimum(rng.zipf(1.1, (3, live)), R - 1)
        keys[:, live:] = 55                             # beyond the live length: must not be read
        n = 3 * stride
        kw = dict(n_dev=torch.tensor([130], dtype=torch.int32, device="cuda"), n_mul=16)
    g = rng.standard_normal((n, D)).astype(np.float32)
    flat = keys.reshape(-1).astype(np.int64)
    if case == "regions_dev":
        use = np.zeros((3, 4096), bool); use[:, :16 * 130] = True
        flat_used, g_used = flat[use.reshape(-1)], g[use.reshape(-1)]
    else:
        flat_used, g_used = flat, g
    ok = (flat_used > 0) & (flat_used < R)
    We, me, ve = W0.copy(), m0.copy(), v0.copy()
    if ok.any():
        oadam.sparse_adam_rows(We, me, ve, flat_used[ok], g_used[ok], 7, 1e-3, wd=1e-2, padding_idx=0)
    outs = []
    for _ in range(2):
        W, m, v = (torch.from_numpy(a.copy()).cuda() for a in (W0, m0, v0))
        ops.sparse_adam_rows_small(torch.from_numpy(g).cuda(), torch.from_numpy(keys).cuda(), W, m, v, step=7, lr=1e-3, weight_decay=1e-2,
                                   padding_idx=0, **kw)
        outs.append((W.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    W, m, v = outs[0]
    touched = np.unique(flat_used[ok])
    tol = dict(rtol=5e-5, atol=5e-5)     # (a Zipf-head row sums thousands of rows: fp32 sums in another association than the oracle's f64-free loop)
    np.testing.assert_allclose(W[touched], We[touched], **tol)
    np.testing.assert_allclose(m[touched], me[touched], **tol)
    np.testing.assert_allclose(v[touched], ve[touched], **tol)
    untouched = np.setdiff1d(np.arange(R), touched)
    assert np.array_equal(W[untouched], W0[untouched]) and np.array_equal(m[untouched], m0[untouched]) and np.array_equal(v[untouched], v0[untouched])
    # the device-scalar form (captured steps) takes the same step
    hyper = torch.tensor([1e-3 / (1 - 0.9 ** 7), 1 / np.sqrt(1 - 0.999 ** 7)], dtype=torch.float32, device="cuda")
    W2, m2, v2 = (torch.from_numpy(a.copy()).cuda() for a in (W0, m0, v0))
    ops.sparse_adam_rows_small(torch.from_numpy(g).cuda(), torch.from_numpy(keys).cuda(), W2, m2, v2, weight_decay=1e-2, padding_idx=0, hyper=hyper, **kw)
    np.testing.assert_allclose(W2.cpu().numpy(), W, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("case", ["long_tiles_at_the_limit", "three_tile_chains", "singles_and_empties", "one_sequence", "maxlen_16", "maxlen_64",
                                  "more_long_tiles_than_workgroups"])
def test_tile_kernel_agrees_with_the_fp32_item_kernels(case, D):
    """The one-tile-per-workgroup step (bf16 split products, hand-over flags between the workgroups of a long sequence) on batch compositions
    that stress the hand-over -- 192 long tiles, chains of three tiles, single-token and empty rows, a batch of one, maxlen 16 (no chains
    at all) and 64 (four full tiles), more long tiles than there are resident workgroups (several passes per workgroup) -- against the CPU ORACLE with the same masks and pinned relu gates (loss
    to 2e-5, every gradient entry to 1e-4 of its tensor's largest), and against the fp32 workgroup-per-item kernels (L2).  The plan must
    have chosen the tile kernel; no hand-over time-out."""
    from recboard_amd.sasrec import SASRecEngine
    rng = np.random.default_rng(17)
    N, L, p = 900, 2, 0.25
    S = {"maxlen_16": 16, "maxlen_64": 64}.get(case, 50)
    if case == "long_tiles_at_the_limit":
        lens = [49] * 48 + list(rng.integers(1, 16, 120))                 # 48 x 4 = 192 long tiles + short ones
    elif case == "more_long_tiles_than_workgroups":
        # 220 x 4 + 90 x 3 = 1 150 long tiles + short ones for 256 resident workgroups: every workgroup makes several passes, chains cross the
        # grid's end (tile 255 | 256 of one sequence are the last and the first workgroup, one pass apart)
        # (+ empty rows: more than 512 sequences -- the looped form of the kernel, enc_common.h: enc_tile_looped)
        lens = [49] * 220 + [int(x) for x in rng.integers(33, 49, 90)] + list(rng.integers(1, 17, 150)) + [0] * 100
    elif case == "three_tile_chains":
        lens = [int(x) for x in rng.integers(33, 49, 64)] + list(rng.integers(1, 17, 215))   # (192 long tiles)
    elif case == "singles_and_empties":
        lens = [1] * 150 + [0] * 20 + [2] * 30 + [17] * 5
    elif case == "one_sequence":
        lens = [37]
    elif case == "maxlen_16":
        lens = list(rng.integers(0, 17, 200))
    else:
        lens = [64] * 40 + [63, 49, 48, 33, 32, 17, 16] + list(rng.integers(1, 16, 60))
    B = len(lens)
    seq = np.zeros((B, S), np.int64)
    for b, n in enumerate(lens):
        n = min(int(n), S)
        if n:
            seq[b, S - n:] = rng.integers(1, N + 1, n)
    pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    batch = tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg))
    from oracle import sasrec as osas
    from recboard_amd import ops
    res = []
    for tile in (True, False):
        m = SASRecEngine(N, S, D, L, dropout_rate=p, loss="BCE", lr=0.0, weight_decay=0.0, seed=6)
        with torch.no_grad():
            g = torch.Generator().manual_seed(3)
            for k, q in m.params.items():
                if k.endswith("bias"):
                    q.copy_((0.05 * torch.randn(q.shape, generator=g)).cuda())
                elif "LN" in k:
                    q.copy_((1.0 + 0.1 * torch.randn(q.shape, generator=g)).cuda())
        m.tile_step = "always" if tile else False   # (off: the fp32 workgroup-per-item kernels, one launch at D = 128, two at D = 64; "always": also
                                                    #  where the plan would hand the batch to them for speed -- more_long_tiles_than_workgroups)
        if D == 64:
            m.fused_item_kernel = tile
        m.split_long = False
        pb = m.prepare_batch(*batch)
        if tile:
            assert int(pb.plan.view(torch.int32)[7]) == 1, "the plan did not choose the tile kernel"
        seed = m._step_seed()
        step0 = m.arena.step
        loss = float(m.train_step(*batch, aux=pb))
        m.check_handover()
        res.append((loss, m.arena.grad.clone(), m, pb, seed))
        if tile and case == "more_long_tiles_than_workgroups":
            # which workgroup takes which tile beyond the grid is decided by a counter at run time: the results must not depend on it
            for _ in range(3):
                m.arena.step = step0                     # (lr = 0: the same parameters; the same step number: the same dropout masks)
                assert float(m.train_step(*batch, aux=pb)) == loss
                assert torch.equal(m.arena.grad, res[-1][1])
    # THE TILE KERNEL AGAINST THE ORACLE, entry by entry at the 1e-4 bound: the same dropout masks, the engine's relu gates where the oracle's own
    # pre-activation is within 2e-5 of zero (oracle/sasrec.py: block) -- a key tile missing from a hand-over, or a stale one, moves whole rows
    # of dK / dV and shows in every gradient tensor at this bound
    lval, grad, m, pb, seed = res[0]
    tape = m._buffers(B, S)["tape"]
    gates = {l: ((ops.sasrec_tape_array(tape, pb.plan, B, S, D, L, "HR", l) > 0).cpu(), 2e-5) for l in range(L)}
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    ref = osas.fit(P, batch[0].cpu(), batch[1].cpu(), batch[2].cpu(), "BCE", L, drop=dict(p=p, seed=seed), gates=gates)
    ref.backward()
    assert abs(lval - ref.item()) <= 2e-5 * abs(ref.item()), (case, lval, ref.item())
    Gv = m.arena.views(grad)
    for k, q in P.items():
        r = q.grad if q.grad is not None else torch.zeros_like(q)
        err = (Gv[k].cpu() - r).abs().max().item()
        assert err <= 1e-4 * r.abs().max().item() + 1e-7, (case, k, err, r.abs().max().item())
    # ... and against the fp32 workgroup-per-item kernels (another arithmetic: a relu gate within rounding of zero may fall on the other side in
    # one of the two, so this comparison is in the L2 sense)
    assert abs(res[0][0] - res[1][0]) <= 2e-5 * abs(res[1][0]), (res[0][0], res[1][0])
    Ga, Gb = res[0][2].arena.views(res[0][1]), res[1][2].arena.views(res[1][1])
    for k in Ga:
        ref = Gb[k]
        dn, rn = float((Ga[k] - ref).norm()), float(ref.norm())
        assert dn <= 1e-2 * rn + 1e-9, (case, k, dn, rn)
        assert float((Ga[k] - ref).abs().max()) <= 0.2 * float(ref.abs().max()) + 1e-7, (case, k)
