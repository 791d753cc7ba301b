"""GPU: the device batch sampler (csrc/sampler.hip, recboard_amd/sampler.py) -- the hand-derived rows of tests/golden/sampler_rows.json,
bit-exact agreement with the numpy restatement (oracle/sampler.py), negatives never in a user's training set and uniform over the rest
(chi-square), the freerec pipe's device path, and the epoch rate of sampler -> fused step against resident batches."""
import json
import os
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _inter(seqs, N):
    from recboard_amd.sampler import DeviceInteractions
    ptr = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(s) for s in seqs], out=ptr[1:])
    return DeviceInteractions(ptr, np.concatenate([np.asarray(s, np.int64) for s in seqs]), N), ptr


def test_rows_match_the_hand_derived_fixture_and_the_oracle_bit_for_bit():
    from oracle import sampler as osm
    from recboard_amd import sampler
    fx = json.load(open(os.path.join(HERE, "golden", "sampler_rows.json")))
    N, S = fx["num_items"], fx["maxlen"]
    train = [s[:-2] for s in fx["seqs"]]                   # leave-one-out: the last two items are the valid / test targets
    inter, ptr = _inter(train, N)
    order = torch.arange(len(train), device="cuda")
    users, seq, pos, neg = sampler.seq_train_sample(inter, order, 0, len(train) + 1, S, seed=5, step=3)
    for u, want in fx["train"].items():
        assert seq[int(u)].tolist() == want["ISeq"] and pos[int(u)].tolist() == want["IPos"]
    assert users.tolist() == list(range(len(train))) + [-1] and int(seq[-1].abs().sum() + pos[-1].abs().sum() + neg[-1].abs().sum()) == 0
    ref = osm.seq_train_sample(ptr, inter.items.cpu().numpy(), order.cpu().numpy(), 0, len(train) + 1, S, N, 5, 3)
    for a, b in zip((users, seq, pos, neg), ref):
        assert np.array_equal(a.cpu().numpy(), b)
    for u, s in enumerate(train):
        live = seq[u] > 0
        assert not np.isin(neg[u][live].cpu().numpy(), s).any() and (neg[u][~live] == 0).all()


def test_negatives_are_uniform_over_the_unseen_items():
    from recboard_amd import sampler
    N, S, U = 64, 50, 400
    rng = np.random.default_rng(1)
    seqs = [rng.choice(N, 20, replace=False) for _ in range(U)]          # every user has seen 20 of the 64 items
    inter, _ = _inter(seqs, N)
    order = torch.arange(U, device="cuda")
    counts = np.zeros(N)
    unseen_mass = np.zeros(N)
    for step in range(40):
        _, seq, _, neg = sampler.seq_train_sample(inter, order, 0, U, S, seed=9, step=step)
        live = (seq > 0).cpu().numpy()
        ng = neg.cpu().numpy()
        for u in range(U):
            assert not np.isin(ng[u][live[u]], seqs[u]).any()
        counts += np.bincount(ng[live], minlength=N)
        per_user = live.sum(1)
        for u in range(U):
            m = np.ones(N, bool); m[seqs[u]] = False
            unseen_mass[m] += per_user[u] / m.sum()
    chi2 = ((counts - unseen_mass) ** 2 / unseen_mass).sum()
    assert chi2 < 63 + 5 * np.sqrt(2 * 63), chi2                        # 63 degrees of freedom: mean 63, sd 11.2


def test_freerec_pipe_runs_on_the_device_and_keeps_the_row_contract():
    import freerec
    from test_freerec_compat import ROOT, import_script
    own = import_script(os.path.join(ROOT, "examples", "SASRec", "main.py"), "_own_sasrec_devpipe", ["--maxlen", "50"])
    rng = np.random.default_rng(2)
    seqs = [rng.integers(0, 500, rng.integers(3, 80)).tolist() for _ in range(300)]
    ds = freerec.data.datasets.RecDataSet.from_sequences(seqs, 500)
    model = own.SASRec(ds)
    pipe = model.sure_trainpipe(50, 64).to_("cuda")
    seen_users = []
    for b in pipe:
        seq, pos, neg, users = b[model.ISeq], b[model.IPos], b[model.INeg], b[model.User]
        assert seq.is_cuda and seq.shape[1] == 50 and b[model.Size] == seq.shape[0]
        for r, u in enumerate(users.tolist()):
            tr = np.asarray(seqs[u][:-2])
            w = tr[-50:]                        # shuffled_seqs_source(maxlen): the last maxlen items, THEN the target is split off
            want_seq = np.zeros(50, np.int64); want_seq[50 - (len(w) - 1):] = w[:-1] + 1
            want_pos = np.zeros(50, np.int64); want_pos[50 - (len(w) - 1):] = w[1:]
            assert np.array_equal(seq[r].cpu().numpy(), want_seq) and np.array_equal(pos[r].cpu().numpy(), want_pos)
            live = want_seq > 0
            assert not np.isin(neg[r].cpu().numpy()[live], tr).any()
        seen_users += users.tolist()
    assert sorted(seen_users) == [u for u in range(300) if len(seqs[u]) - 2 >= 2]       # every eligible user exactly once per epoch


def test_sampler_to_step_epoch_rate_is_at_least_0p9_of_resident_batches():
    from recboard_amd import sampler
    from recboard_amd.sasrec import SASRecEngine
    N, B, S, U = 12101, 512, 50, 22363
    rng = np.random.default_rng(3)
    w = 1.0 / np.arange(1, N + 1); w /= w.sum()
    lens = np.clip(rng.geometric(1 / 5.9, U) + 2, 3, 60)
    seqs = [rng.choice(N, n, p=w) for n in lens]
    inter, _ = _inter(seqs, N)
    eng = SASRecEngine(N, S, 64, 2, dropout_rate=0.5, loss="BCE", lr=5e-4, weight_decay=1e-6, seed=1)
    smp = sampler.DeviceSeqSampler(inter, S, B, seed=1)
    resident = [(b["ISeq"], b["IPos"], b["INeg"]) for b in smp if b["ISeq"].shape[0] == B]

    def epoch_resident():
        for seq, pos, neg in resident:
            eng.train_step_graph(seq, pos, neg)
        torch.cuda.synchronize()

    def epoch_sampled():
        for b in smp:
            if b["ISeq"].shape[0] == B:
                eng.train_step_graph(b["ISeq"], b["IPos"], b["INeg"])
        torch.cuda.synchronize()
    best = {}
    for name, fn in (("resident", epoch_resident), ("sampled", epoch_sampled)) * 3:
        t0 = time.perf_counter()
        fn()
        best[name] = min(best.get(name, 1e9), time.perf_counter() - t0)
    assert best["resident"] >= 0.9 * best["sampled"], best


def test_sample_and_prepare_in_one_launch_equals_sampler_then_prepare():
    """re_seq_train_sample_prep: the staging blob it leaves (batch, valid mask, count, destination rows, plan) is bit for bit what
    re_seq_train_sample followed by re_sasrec_batch_prep leave; a training epoch through tickets equals the epoch through tensors."""
    from recboard_amd import ops
    from recboard_amd.coach import Coach
    from recboard_amd.sampler import DeviceInteractions, DeviceSeqSampler, seq_train_sample
    from recboard_amd.sasrec import SASRecEngine
    rng = np.random.default_rng(21)
    U, N, S, B = 700, 400, 50, 96
    lens = np.concatenate([rng.integers(2, 12, U - 40), rng.integers(30, 120, 40)])
    ptr = np.zeros(U + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    items = np.concatenate([rng.choice(N, l, replace=False) if l <= N else rng.integers(0, N, l) for l in lens])
    inter = DeviceInteractions(ptr, items, N)
    order = inter.users_ge2[torch.randperm(inter.users_ge2.numel(), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))]
    for b0, Bq in ((0, B), (U - 28, 28)):                                       # a full batch and the epoch's short last one
        users, seq, pos, neg = seq_train_sample(inter, order, b0, Bq, S, 7, 3)
        ref = ops.sasrec_batch_prep(seq, pos, neg, blob=torch.zeros(ops.prep_layout(Bq, S)[1], dtype=torch.uint8, device="cuda"))
        blob = torch.zeros_like(ref.blob)
        u2 = torch.empty(Bq, dtype=torch.int64, device="cuda")
        got = ops.sasrec_sample_prep(inter, order, b0, Bq, S, 7, 3, blob, users=u2)
        torch.cuda.synchronize()
        assert torch.equal(u2, users)
        for name in ("seq", "pos", "neg", "rows_all", "valid", "count"):
            assert torch.equal(getattr(got, name), getattr(ref, name)), name
        nw = 8 + 2 * int(got.plan.view(torch.int32)[0])                         # header + item words; the row map is compared below
        assert torch.equal(got.plan.view(torch.int32)[:8], ref.plan.view(torch.int32)[:8])
        assert torch.equal(got.plan, ref.plan), nw
    # an epoch either way: the same losses and parameters (tickets carry the same (order, seed, step) the tensor batches were made from)
    res = []
    for fused in (False, True):
        m = SASRecEngine(N, S, 64, 2, dropout_rate=0.2, lr=1e-3, seed=5)
        smp = DeviceSeqSampler(inter, S, B, seed=9, fused=fused)
        coach = Coach(m, smp, monitors=["LOSS"], kind="seq")
        res.append((coach.train_per_epoch(0)["LOSS"], coach.train_per_epoch(1)["LOSS"], m.arena.data.clone()))
        m.check_handover()
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[0][0]) and abs(res[0][1] - res[1][1]) <= 1e-6 * abs(res[0][1])
    torch.testing.assert_close(res[1][2], res[0][2], rtol=1e-5, atol=1e-7)
    # the Coach hands every step the NEXT ticket (sampled + prepared by jobs of the step's tail launch): the same epoch, bit for bit, as with
    # every ticket sampled by a launch in front of its step
    m = SASRecEngine(N, S, 64, 2, dropout_rate=0.2, lr=1e-3, seed=5)
    m.prep_in_tail = False
    coach = Coach(m, DeviceSeqSampler(inter, S, B, seed=9, fused=True), monitors=["LOSS"], kind="seq")
    l0, l1 = coach.train_per_epoch(0)["LOSS"], coach.train_per_epoch(1)["LOSS"]
    assert (l0, l1) == (res[1][0], res[1][1])
    assert torch.equal(m.arena.data, res[1][2])
    assert not getattr(m, "_tail_pipes", {})
