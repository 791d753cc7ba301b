"""GPU: owner-bucketing routing kernel, exact AUC kernel, the prediction Coach (DeepFM: LOGLOSS / AUC, ReduceLROnPlateau, checkpoint
round trip), and the large-table engine through the Coach (evaluate + checkpoint with the table's Adam moments)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.mark.parametrize("n,R,G,factor,skip", [(76800, 100_000_001, 8, 2.0, -1), (5000, 977, 3, 1.5, -1), (1, 10, 2, 1.0, -1), (3000, 50, 8, 0.5, -1),
                                               (0, 10, 4, 1.0, -1), (76800, 100_000_001, 8, 0.4, 0), (5000, 977, 3, 1.5, 1)])
def test_route_bucket_matches_numpy_restatement(n, R, G, factor, skip):
    """re_route_bucket (csrc/route.hip): stable counting sort by owner (row mod G) into fixed-capacity buckets; overflow and
    out-of-range lookups are counted, never written."""
    _gpu()
    from recboard_amd import ops
    rng = np.random.default_rng(n + G)
    idx = np.minimum(rng.zipf(1.05, n), R - 1).astype(np.int64) if n else np.zeros(0, np.int64)
    if n > 10:
        idx[3], idx[7] = -5, R + 2                       # out of range: dropped
    if skip >= 0 and n > 10:
        idx[rng.random(n) < 0.8] = skip                  # the padding row: most of a left-padded batch; takes no slot
    cap = max(1, min(max(n, 1), int(np.ceil(factor * max(n, 1) / G))))
    buckets = np.full((G, cap), -1, np.int64)
    slot = np.full(n, -1, np.int64)
    counts = np.zeros(G + 1, np.int64)
    for j, r in enumerate(idx.tolist()):
        if r == skip:
            continue
        if r < 0 or r >= R:
            counts[G] += 1
            continue
        g = r % G
        k = counts[g]
        counts[g] += 1
        if k < cap:
            buckets[g, k], slot[j] = r // G, g * cap + k
        else:
            counts[G] += 1
    b, s, c = ops.route_bucket(torch.from_numpy(idx).cuda(), R, G, cap, skip_row=skip)
    np.testing.assert_array_equal(c.cpu().numpy(), counts)
    np.testing.assert_array_equal(b.cpu().numpy(), buckets)
    np.testing.assert_array_equal(s.cpu().numpy(), slot)


@pytest.mark.parametrize("n,ties", [(5000, False), (30000, True), (257, True), (2, False)])
def test_auc_kernel_is_the_exact_mann_whitney_statistic(n, ties):
    _gpu()
    from sklearn.metrics import roc_auc_score
    from recboard_amd import ops
    rng = np.random.default_rng(n)
    y = (rng.random(n) < 0.3).astype(np.float32)
    y[0], y[-1] = 1.0, 0.0
    s = (rng.standard_normal(n) + 0.7 * y).astype(np.float32)
    if ties:
        s = np.round(s * 4) / 4                          # heavy ties: each tie counts one half
    got = float(ops.auc(torch.from_numpy(s).cuda(), torch.from_numpy(y).cuda()))
    assert abs(got - roc_auc_score(y, s)) < 1e-6
    assert float(ops.auc(torch.from_numpy(s).cuda(), torch.ones(n, device="cuda"))) == 0.5      # an empty class


def test_prediction_coach_deepfm_logloss_auc_plateau_scheduler_and_checkpoint(tmp_path):
    """CoachForDeepFM's loop (DeepFM/main.py:251-276) on the engine: LOGLOSS / AUC monitors, ReduceLROnPlateau stepped on the best value
    at the top of every epoch, checkpoint.tar with the reference's keys; a restored engine (BatchNorm statistics included) evaluates identically."""
    _gpu()
    from sklearn.metrics import log_loss, roc_auc_score
    from recboard_amd.coach import Coach
    from recboard_amd.deepfm import DeepFMEngine
    from recboard_amd.evaluate import ReduceLROnPlateau
    counts, B = [50, 40, 7, 3], 512
    rng = np.random.default_rng(0)
    w = [rng.standard_normal(c) for c in counts]

    def make(nb, seed):
        r = np.random.default_rng(seed)
        out = []
        for _ in range(nb):
            x = np.stack([r.integers(0, c, B) for c in counts], 1)
            logit = sum(w[f][x[:, f]] for f in range(len(counts)))
            y = (r.random(B) < 1 / (1 + np.exp(-logit))).astype(np.float32)
            out.append({"X": torch.from_numpy(x), "Label": torch.from_numpy(y).reshape(B, 1)})
        return out

    train, valid = make(12, 1), make(4, 2)
    m = DeepFMEngine(counts, 10, (64, 32), batch_norm=True, lr=5e-3, embedding_decay=0.0, seed=1)
    sched = ReduceLROnPlateau(m, mode="max", patience=1, factor=0.1, threshold=1e-6, min_lr=1e-6)
    coach = Coach(m, train, valid, monitors=["LOSS", "LOGLOSS", "AUC"], which4best="AUC", eval_freq=1, kind="pred",
                  checkpoint_path=str(tmp_path), lr_scheduler=sched)
    before = coach.evaluate("valid")
    out = coach.fit(6)
    after = out["history"][-1]["valid"]
    assert abs(before["AUC"] - 0.5) < 0.1 and after["AUC"] > 0.7 and after["LOGLOSS"] < before["LOGLOSS"]
    # the metrics are what sklearn computes from the same probabilities
    m.eval()
    z = torch.cat([m.encode(b["X"].cuda())[0] for b in valid]).cpu().numpy()
    y = np.concatenate([b["Label"].numpy().reshape(-1) for b in valid])
    p = 1 / (1 + np.exp(-z.astype(np.float64)))
    assert abs(after["AUC"] - roc_auc_score(y, p)) < 1e-5 and abs(after["LOGLOSS"] - log_loss(y, p)) < 1e-4
    # scheduler: torch's ReduceLROnPlateau fed the same sequence of best values gives the same learning rates
    ref_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=5e-3)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(ref_opt, mode="max", patience=1, factor=0.1, threshold=1e-6, min_lr=1e-6)
    mine = ReduceLROnPlateau(type("M", (), {"lr": 5e-3})(), mode="max", patience=1, factor=0.1, threshold=1e-6, min_lr=1e-6)
    for best in [-float("inf"), 0.6, 0.7, 0.7, 0.7, 0.7, 0.71, 0.71, 0.71, 0.71, 0.71]:
        ref.step(best); mine.step(best)
        assert abs(ref_opt.param_groups[0]["lr"] - mine.model.lr) < 1e-12
    assert mine.model.lr < 5e-3
    ck = torch.load(tmp_path / "checkpoint.tar", weights_only=False)
    assert set(ck) == {"epoch", "model", "optimizer", "lr_scheduler", "monitors"}
    m2 = DeepFMEngine(counts, 10, (64, 32), batch_norm=True, lr=5e-3, embedding_decay=0.0, seed=9)
    coach2 = Coach(m2, None, valid, monitors=["LOGLOSS", "AUC"], which4best="AUC", kind="pred", lr_scheduler=ReduceLROnPlateau(m2, patience=1))
    assert coach2.load_checkpoint(str(tmp_path)) == 6
    assert coach2.evaluate("valid") == after and m2.step == m.step and coach2.lr_scheduler.best == sched.best


def test_large_table_engine_through_the_coach_evaluate_and_checkpoint(tmp_path):
    """The config-5 engine (item table outside the arena) driven by the Coach: evaluate() (reset_ranking_buffers included) and a
    checkpoint round trip that keeps the table's Adam moments -- the next step after a restore equals the next step without one."""
    _gpu()
    from recboard_amd.coach import Coach
    from recboard_amd.data import EvalSampler, SeqTrainSampler, SyntheticSeqDataset
    from recboard_amd.large import SASRecLargeTableEngine
    ds = SyntheticSeqDataset(600, 300, mean_len=9, p_follow=0.8, seed=3)
    kw = dict(dropout_rate=0.0, loss="BCE", lr=1e-3, weight_decay=1e-5, seed=2)
    m = SASRecLargeTableEngine(300, 50, 128, 2, **kw)
    coach = Coach(m, SeqTrainSampler(ds, 50, 128, seed=1), EvalSampler(ds, 50, 256, "valid"), monitors=["LOSS", "HitRate@10", "NDCG@10"],
                  eval_freq=2, kind="seq", checkpoint_path=str(tmp_path))
    out = coach.fit(4)
    assert out["history"][-1]["valid"]["HITRATE@10"] > 0.15             # learns the planted transitions (untrained: ~0.03)
    m2 = SASRecLargeTableEngine(300, 50, 128, 2, **dict(kw, seed=5))
    coach2 = Coach(m2, None, EvalSampler(ds, 50, 256, "valid"), monitors=["HitRate@10", "NDCG@10"], kind="seq")
    assert coach2.load_checkpoint(str(tmp_path)) == 4
    assert coach2.evaluate("valid") == coach.evaluate("valid")
    assert torch.equal(m2.Em, m.Em) and torch.equal(m2.arena.m, m.arena.m) and m2.arena.step == m.arena.step
    batch = next(iter(SeqTrainSampler(ds, 50, 64, seed=9)))
    args = [batch[k].cuda() for k in ("ISeq", "IPos", "INeg")]
    la, lb = m.train_step(*args), m2.train_step(*args)
    assert torch.equal(la, lb) and torch.equal(m.E, m2.E)
