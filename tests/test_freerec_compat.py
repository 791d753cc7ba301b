"""The `freerec`-compatible surface (freerec/: the builder's own package) against the reference's model files IMPORTED IN PLACE from
/root/reference (never copied; skipped where the reference is absent, i.e. on the GPU box): the scripts import, their models build on
this package's datasets / fields, load the golden state dicts, and reproduce the golden losses, gradients and scores -- which were
produced with the test stand-in (tests/golden/_freerec_standin.py), so the two restatements of the FreeRec surface pin each other."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is not on this machine")


def import_script(path, name, argv):
    """Import a model script under a private module name (its module-level `cfg.compile()` parses sys.argv)."""
    old_argv, old_path = sys.argv, list(sys.path)
    sys.argv = ["main.py"] + argv
    sys.path.insert(0, os.path.dirname(path))
    sys.path.insert(0, ROOT)
    for k in [k for k in sys.modules if k == "freerec" or k.startswith("freerec.")]:
        if "golden" in str(getattr(sys.modules[k], "__file__", "")):
            del sys.modules[k]
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        sys.argv, sys.path[:] = old_argv, old_path


def toy_dataset(U, N):
    import freerec
    e = np.zeros(0, np.int64)
    return freerec.data.datasets.RecDataSet.from_splits((e, e), (e, e), (e, e), U, N)


def load_golden_state(model, z):
    sd = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}
    model.load_state_dict({k: sd[k] for k in model.state_dict() if k in sd}, strict=False)
    assert set(k for k, _ in model.named_parameters()) <= set(sd)


@needs_ref
@pytest.mark.parametrize("loss,fixture", [("BCE", "sasrec_bce.npz"), ("BPR", "sasrec_bpr.npz"), ("CE", "sasrec_ce.npz")])
def test_reference_sasrec_runs_unchanged_and_reproduces_the_golden(loss, fixture):
    ref = import_script(os.path.join(REF, "SASRec", "main.py"), f"_ref_sasrec_{loss}", ["--dropout-rate", "0", "--loss", loss])
    z = np.load(os.path.join(G, fixture))
    model = ref.SASRec(toy_dataset(40, int(z["cfg/N"])))
    load_golden_state(model, z)
    data = {model.ISeq: torch.from_numpy(z["in/seq"]), model.IPos: torch.from_numpy(z["in/pos"]), model.INeg: torch.from_numpy(z["in/neg"])}
    model.train()
    out = model(data)
    out["rec_loss"].backward()
    np.testing.assert_allclose(out["rec_loss"].item(), float(z["out/rec_loss"]), rtol=1e-6)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), z["grad/" + k], rtol=1e-5, atol=1e-8, err_msg=k)
    model.eval()
    with torch.no_grad():
        np.testing.assert_allclose(model(data, ranking="full").numpy(), z["out/scores"], rtol=1e-5, atol=2e-6)


@needs_ref
def test_reference_mfbpr_and_lightgcn_run_unchanged():
    z = np.load(os.path.join(G, "mfbpr.npz"))
    ref = import_script(os.path.join(REF, "MF-BPR", "main.py"), "_ref_mfbpr", [])
    U, N = z["param/User.embeddings.weight"].shape[0], z["param/Item.embeddings.weight"].shape[0]
    model = ref.MF(toy_dataset(U, N))
    load_golden_state(model, z)
    data = {model.User: torch.from_numpy(z["in/users"]), model.IPos: torch.from_numpy(z["in/pos"]), model.INeg: torch.from_numpy(z["in/neg"])}
    model.train()
    out = model(data)
    np.testing.assert_allclose(out["rec_loss"].item(), float(z["out/rec_loss"]), rtol=1e-6)
    # LightGCN: the script's `dataset.train().to_normalized_adj("sym")` on this package's dataset == the golden's adjacency
    zl = np.load(os.path.join(G, "lightgcn.npz"))
    import freerec
    U, N = zl["param/User.embeddings.weight"].shape[0], zl["param/Item.embeddings.weight"].shape[0]
    crow, col = zl["in/adj_crow"], zl["in/adj_col"]                       # the golden's adjacency: user rows hold the (user, item) edges
    rows = np.repeat(np.arange(len(crow) - 1), np.diff(crow))
    eu, ei = rows[rows < U], col[rows < U] - U
    e = np.zeros(0, np.int64)
    ds = freerec.data.datasets.RecDataSet.from_splits((eu, ei), (e, e), (e, e), U, N)
    refl = import_script(os.path.join(REF, "LightGCN", "main.py"), "_ref_lightgcn", [])
    model = refl.LightGCN(ds)
    A = model.Adj.to_dense().numpy()
    ref_adj = torch.sparse_csr_tensor(torch.from_numpy(zl["in/adj_crow"]), torch.from_numpy(zl["in/adj_col"]), torch.from_numpy(zl["in/adj_val"]),
                                      size=A.shape).to_dense().numpy()
    np.testing.assert_allclose(A, ref_adj, rtol=1e-6, atol=1e-8)
    load_golden_state(model, zl)
    data = {model.User: torch.from_numpy(zl["in/users"]), model.IPos: torch.from_numpy(zl["in/pos"]), model.INeg: torch.from_numpy(zl["in/neg"])}
    model.train()
    out = model(data)
    np.testing.assert_allclose(out["rec_loss"].item(), float(zl["out/rec_loss"]), rtol=1e-5)
    np.testing.assert_allclose(out["emb_loss"].item(), float(zl["out/emb_loss"]), rtol=1e-5)


@needs_ref
def test_reference_sasrec_main_fits_end_to_end_on_a_synthetic_dataset(tmp_path):
    """`main()`'s wiring -- model, the three pipes, CoachForSASRec(...).fit() -- on CPU with this package's Coach (two epochs)."""
    import freerec
    ref = import_script(os.path.join(REF, "SASRec", "main.py"), "_ref_sasrec_fit", ["--dropout-rate", "0.2"])
    rng = np.random.default_rng(0)
    perm = rng.permutation(300)
    seqs = []
    for _ in range(150):
        s = [int(rng.integers(0, 300))]
        for _ in range(int(rng.integers(4, 30))):
            s.append(int(perm[s[-1]]) if rng.random() < 0.8 else int(rng.integers(0, 300)))
        seqs.append(s)
    ds = freerec.data.datasets.RecDataSet.from_sequences(seqs, 300)
    cfg = ref.cfg
    cfg.epochs, cfg.eval_freq, cfg.device, cfg.batch_size = 2, 1, "cpu", 32
    cfg.monitors, cfg.which4best, cfg.checkpoint_path = ["LOSS", "HitRate@10", "NDCG@10"], "NDCG@10", str(tmp_path)
    model = ref.SASRec(ds)
    pipe = model.sure_trainpipe(cfg.maxlen, cfg.batch_size)
    b = next(iter(pipe))
    seq, pos, neg = b[model.ISeq], b[model.IPos], b[model.INeg]
    assert seq.shape == (32, 50)
    assert ((seq > 0) | (pos == 0)).all() and ((seq > 0) | (neg == 0)).all()          # pads line up
    coach = ref.CoachForSASRec(dataset=ds, trainpipe=pipe, validpipe=model.sure_validpipe(cfg.maxlen, ranking="full"),
                               testpipe=model.sure_testpipe(cfg.maxlen, ranking="full"), model=model, cfg=cfg)
    out = coach.fit()
    assert set(out["test"]) == {"HITRATE@10", "NDCG@10"} and 0.0 <= out["test"]["NDCG@10"] <= out["test"]["HITRATE@10"] <= 1.0
    assert os.path.exists(os.path.join(str(tmp_path), "checkpoint.tar")) and os.path.exists(os.path.join(str(tmp_path), "results.json"))
    ck = torch.load(os.path.join(str(tmp_path), "checkpoint.tar"), weights_only=False)
    assert set(ck) == {"epoch", "model", "optimizer", "lr_scheduler", "monitors"} and "Item.embeddings.weight" in ck["model"]


def test_own_sasrec_script_has_the_reference_state_dict_names():
    """examples/SASRec/main.py (the builder's model file on this surface) builds the reference's parameter set: the golden state dict
    loads strictly and the golden loss / scores come out (CPU, the script's own torch code)."""
    own = import_script(os.path.join(ROOT, "examples", "SASRec", "main.py"), "_own_sasrec", ["--dropout-rate", "0", "--loss", "BCE"])
    z = np.load(os.path.join(G, "sasrec_bce.npz"))
    model = own.SASRec(toy_dataset(40, int(z["cfg/N"])))
    sd = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}
    model.load_state_dict(sd, strict=True)
    data = {model.ISeq: torch.from_numpy(z["in/seq"]), model.IPos: torch.from_numpy(z["in/pos"]), model.INeg: torch.from_numpy(z["in/neg"])}
    model.train()
    np.testing.assert_allclose(model(data)["rec_loss"].item(), float(z["out/rec_loss"]), rtol=1e-5)
    model.eval()
    with torch.no_grad():
        np.testing.assert_allclose(model(data, ranking="full").numpy(), z["out/scores"], rtol=1e-4, atol=1e-5)


def test_pipes_follow_the_row_contract():
    """tests/golden/sampler_rows.json (rows worked out by hand from HSTU/sampler.py:47-125 / SASRec/main.py:143-157) through the chained
    pipes of this package: train rows (ISeq = seq[:-1] + 1 left-padded, IPos = seq[1:], negatives unseen), valid / test rows."""
    import json
    import freerec
    fx = json.load(open(os.path.join(G, "sampler_rows.json")))
    seqs, N, S = fx["seqs"], fx["num_items"], fx["maxlen"]
    ds = freerec.data.datasets.RecDataSet.from_sequences(seqs, N)
    own = import_script(os.path.join(ROOT, "examples", "SASRec", "main.py"), "_own_sasrec_pipes", ["--maxlen", str(S)])
    model = own.SASRec(ds)
    rows = {}
    for b in model.sure_trainpipe(S, 4):
        for i, u in enumerate(b[model.User].tolist()):
            rows[u] = (b[model.ISeq][i].tolist(), b[model.IPos][i].tolist(), b[model.INeg][i].tolist())
    for u, want in fx["train"].items():
        iseq, ipos, ineg = rows[int(u)]
        assert iseq == want["ISeq"] and ipos == want["IPos"]
        seen = set(seqs[int(u)][:-2])
        assert all((n == 0) if s == 0 else (n not in seen and 0 <= n < N) for s, n in zip(iseq, ineg))
    got = {}
    for b in model.sure_validpipe(S, ranking="full", batch_size=3):
        for i, u in enumerate(b[model.User].tolist()):
            got[u] = (b[model.ISeq][i].tolist(), list(b[model.IUnseen][i]), list(b[model.ISeen][i]))
    for u, want in fx["valid"].items():
        assert got[int(u)][0] == want["ISeq"] and got[int(u)][1] == want["IUnseen"] and got[int(u)][2] == want["ISeen"]
    got = {}
    for b in model.sure_testpipe(S, ranking="full", batch_size=3):
        for i, u in enumerate(b[model.User].tolist()):
            got[u] = (b[model.ISeq][i].tolist(), list(b[model.IUnseen][i]), list(b[model.ISeen][i]))
    for u, want in fx["test"].items():
        assert got[int(u)][0] == want["ISeq"] and got[int(u)][1] == want["IUnseen"] and got[int(u)][2] == want["ISeen"]


def test_next_item_eval_pipe_emits_one_row_per_held_out_item_on_ratio_splits():
    """A ratio split (several held-out items per user): the sequence evaluation pipe emits one row per held-out item with a growing history
    (seq = seen + unseen[:k], one target, `seen` unchanged: HSTU/sampler.py:101-122); the general (user-id) pipe keeps one row per user."""
    import freerec
    tr = (np.array([0, 0, 0, 1, 1]), np.array([5, 6, 7, 1, 2]))
    va = (np.array([0, 0, 1]), np.array([8, 9, 3]))
    e = np.zeros(0, np.int64)
    ds = freerec.data.datasets.RecDataSet.from_splits(tr, va, (e, e), 2, 12)
    own = import_script(os.path.join(ROOT, "examples", "SASRec", "main.py"), "_own_sasrec_rou", ["--maxlen", "6"])
    model = own.SASRec(ds)
    rows = []
    for b in model.sure_validpipe(6, ranking="full", batch_size=8):
        for i in range(len(b[model.User])):
            rows.append((int(b[model.User][i]), b[model.ISeq][i].tolist(), list(b[model.IUnseen][i]), sorted(b[model.ISeen][i])))
    assert rows == [(0, [0, 0, 0, 6, 7, 8], [8], [5, 6, 7]), (0, [0, 0, 6, 7, 8, 9], [9], [5, 6, 7]), (1, [0, 0, 0, 0, 2, 3], [3], [1, 2])]
    mf = import_script(os.path.join(ROOT, "examples", "MF-BPR", "main.py"), "_own_mf_rou", [])
    gen = mf.MF(ds)
    got = [(int(b[gen.User][i]), list(b[gen.IUnseen][i])) for b in gen.sure_validpipe("full") for i in range(len(b[gen.User]))]
    assert got == [(0, [8, 9]), (1, [3])]


def _aggregate_like_build_data(evaluations):
    """recboard/scripts/build-data.mjs:49-82,116-131 restated: normalizeRun + aggregateRuns over a file's evaluations."""
    out = []
    for ev in evaluations if isinstance(evaluations, list) else [evaluations]:
        runs = [dict(r, metrics={k: (r.get("metrics") or {}).get(k) or {} for k in ("train", "valid", "test", "best")}) for r in ev.get("runs") or []]
        keys = list(runs[0]["metrics"]["best"]) if runs else []
        best = {}
        for k in keys:
            vals = [r["metrics"]["best"][k] for r in runs if r["metrics"]["best"].get(k) is not None]
            mean = sum(vals) / len(vals) if vals else 0.0
            best[k] = {"mean": mean, "std": (sum((v - mean) ** 2 for v in vals) / len(vals)) ** 0.5 if vals else 0.0}
        out.append({"description": ev.get("description") or "", "tags": ev.get("tags") or [], "bestMetrics": best, "runs": runs,
                    "config": ev.get("config") or {}, "timestamp": ev.get("timestamp") or ""})
    return out


def test_fit_of_an_unchanged_script_writes_the_leaderboard_record(tmp_path):
    """`Coach.fit()` of examples/SASRec/main.py (CPU: the script's own torch code) ends with `results.json` in the
    schema of benchmark/<dataset>/<model>.json (benchmark/Amazon2014Beauty_550_LOU/SASRec.json:1-304), field by field the shape of the
    published rows (tests/golden/benchmark_rows.json) and readable by the restated aggregation of recboard/scripts/build-data.mjs:95-146."""
    import datetime
    import json
    import re
    import freerec
    own = import_script(os.path.join(ROOT, "examples", "SASRec", "main.py"), "_own_sasrec_record", ["--dropout-rate", "0.2", "--seed", "3"])
    rng = np.random.default_rng(0)
    perm = rng.permutation(200)
    seqs = []
    for _ in range(120):
        s = [int(rng.integers(0, 200))]
        for _ in range(int(rng.integers(4, 20))):
            s.append(int(perm[s[-1]]) if rng.random() < 0.8 else int(rng.integers(0, 200)))
        seqs.append(s)
    ds = freerec.data.datasets.RecDataSet.from_sequences(seqs, 200)
    cfg = own.cfg
    cfg.epochs, cfg.eval_freq, cfg.device, cfg.batch_size, cfg.dataset = 3, 1, "cpu", 32, "Synthetic_550_LOU"
    published = json.load(open(os.path.join(G, "benchmark_rows.json")))["Amazon2014Beauty_550_LOU/SASRec"][0]
    cfg.monitors = ["LOSS"] + sorted(published["valid"])          # the published run's monitors: HITRATE@{1,5,10,20,50}, NDCG@{5,10,20,50}
    cfg.which4best, cfg.checkpoint_path, cfg.eval_test = "NDCG@10", str(tmp_path), True
    model = own.SASRec(ds)
    coach = own.CoachForSASRec(dataset=ds, trainpipe=model.sure_trainpipe(cfg.maxlen, cfg.batch_size), validpipe=model.sure_validpipe(cfg.maxlen, ranking="full"),
                               testpipe=model.sure_testpipe(cfg.maxlen, ranking="full"), model=model, cfg=cfg)
    out = coach.fit()
    rec = json.load(open(tmp_path / "results.json"))
    assert isinstance(rec, list) and len(rec) == 1
    ev = rec[0]
    assert list(ev) == ["description", "dataset", "tags", "runs", "timestamp", "config"]       # the published file's keys, in its order
    assert ev["dataset"] == "Synthetic_550_LOU" and isinstance(ev["description"], str) and all(isinstance(t, str) for t in ev["tags"]) and ev["tags"]
    datetime.datetime.strptime(ev["timestamp"], "%Y-%m-%dT%H:%M:%S")
    assert len(ev["runs"]) == 1
    run = ev["runs"][0]
    assert list(run) == ["id", "params", "metrics"] and isinstance(run["id"], str) and set(run["params"]) == {"config", "seed"} and run["params"]["seed"] == 3
    assert list(run["metrics"]) == ["train", "valid", "test", "best"]
    # field by field the published row's shape: the same metric names per split, plain floats
    for split in ("train", "valid", "test", "best"):
        assert set(run["metrics"][split]) == set(published[split]), (split, sorted(run["metrics"][split]))
        assert all(isinstance(v, float) and 0.0 <= v for v in run["metrics"][split].values())
    assert all(re.fullmatch(r"(HITRATE|NDCG)@\d+", k) for k in run["metrics"]["best"])
    m = run["metrics"]
    assert m["valid"] == out["valid"] and m["test"] == out["test"] and m["best"] == out["best_test"] and m["train"] == out["history"][-2]["train"]
    for split in ("valid", "test", "best"):       # the identities the published rows satisfy (one held-out target per user)
        assert m[split]["HITRATE@1"] <= m[split]["HITRATE@5"] <= m[split]["HITRATE@10"] <= m[split]["HITRATE@50"]
        assert m[split]["NDCG@10"] <= m[split]["HITRATE@10"] + 1e-12
    # the config dump: what the script's cfg holds, JSON-clean (build-data.mjs filters its own blacklist)
    assert ev["config"]["maxlen"] == cfg.maxlen and ev["config"]["which4best"] == "NDCG@10" and ev["config"]["epochs"] == 3
    agg = _aggregate_like_build_data(rec)[0]
    assert set(agg["bestMetrics"]) == set(published["best"]) and agg["bestMetrics"]["NDCG@10"]["mean"] == m["best"]["NDCG@10"] and agg["bestMetrics"]["NDCG@10"]["std"] == 0.0
    # history: one record per epoch (evaluation in front of the epoch) + the final evaluations; the best checkpoint exists
    assert [h["epoch"] for h in out["history"]] == [0, 1, 2, 3] and all("train" in h for h in out["history"][:-1]) and "valid" in out["history"][-1]
    assert os.path.exists(tmp_path / "best.pt")
