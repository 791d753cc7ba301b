"""CPU: host logic of the vectorised samplers / evaluator helpers (row contracts of the reference's datapipes)."""
import numpy as np
import torch

from recboard_amd.data import EvalSampler, GenTrainSampler, SeqTrainSampler, SyntheticSeqDataset
from recboard_amd.evaluate import parse_monitors, ragged_to_csr


def test_seq_train_rows_follow_the_reference_contract():
    ds = SyntheticSeqDataset(200, 60, mean_len=9, seed=2)
    sp = SeqTrainSampler(ds, maxlen=50, batch_size=64, seed=1)
    seen_users = []
    for batch in sp:
        for r in range(len(batch["User"])):
            u = int(batch["User"][r])
            tr = ds.train_seq(u)[-51:]
            iseq, ipos, ineg = batch["ISeq"][r].numpy(), batch["IPos"][r].numpy(), batch["INeg"][r].numpy()
            L = len(tr) - 1
            assert (iseq[:50 - L] == 0).all() and (ipos[:50 - L] == 0).all() and (ineg[:50 - L] == 0).all()   # lpad_ with 0
            np.testing.assert_array_equal(iseq[50 - L:], tr[:-1] + 1)       # seq[:-1], NUM_PADS offset on ISeq only
            np.testing.assert_array_equal(ipos[50 - L:], tr[1:])            # seq[1:], 0-based
            assert not set(ineg[50 - L:].tolist()) & set(ds.train_seq(u).tolist())   # negatives are unseen items
            seen_users.append(u)
    assert sorted(seen_users) == sorted(sp.users.tolist())                  # one row per user per epoch


def test_gen_train_and_eval_rows():
    ds = SyntheticSeqDataset(100, 80, mean_len=8, seed=4)
    gp = GenTrainSampler(ds, 128, seed=1)
    b = next(iter(gp))
    assert b["User"].shape == (128, 1) and b["IPos"].shape == (128, 1) and b["INeg"].shape == (128, 1)
    for u, p, n in zip(b["User"][:, 0].tolist(), b["IPos"][:, 0].tolist(), b["INeg"][:, 0].tolist()):
        assert p in ds.train_seq(u) and n not in ds.train_seq(u)
    assert len(gp) == (ds.num_train_interactions() + 127) // 128
    ev = next(iter(EvalSampler(ds, 50, 32, "test")))
    u = int(ev["User"][3])
    assert ev["IUnseen"][3] == [int(ds.test_target(u))]
    assert ev["ISeen"][3] == sorted(set(ds.seqs[u][:-1].tolist()))          # test: train + valid items are seen
    assert int(ev["ISeq"][3, -1]) == int(ds.seqs[u][-2]) + 1                # last position is the most recent item


def test_evaluator_helpers():
    assert parse_monitors(["LOSS", "HitRate@10", "ndcg@5"]) == [("HITRATE", 10), ("NDCG", 5)]
    ptr, idx = ragged_to_csr([[5, 1], [], [3]], "cpu")
    assert ptr.tolist() == [0, 2, 2, 3] and idx.tolist() == [1, 5, 3]


def test_sampler_rows_match_the_reference_row_contract_fixture():
    """tests/golden/sampler_rows.json: rows derived BY HAND from the reference's samplers for a three-user toy data set
    (HSTU/sampler.py:54-62 train: IPos = seq[1:], ISeq = seq[:-1]; :107-125 valid rows: seq = train items, IUnseen = (valid item,),
    ISeen = train items; :131-172 test rows: seen = train + valid; SASRec/main.py:143-157: ISeq + NUM_PADS, lprune/lpad to maxlen)."""
    import json
    import os
    from recboard_amd.data import ExplicitSeqDataset
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sampler_rows.json")))
    ds = ExplicitSeqDataset(fx["seqs"], fx["num_items"])
    S = fx["maxlen"]
    got = {}
    for b in SeqTrainSampler(ds, S, 8, seed=0):
        for r, u in enumerate(b["User"].tolist()):
            got[str(u)] = b
            assert b["ISeq"][r].tolist() == fx["train"][str(u)]["ISeq"] and b["IPos"][r].tolist() == fx["train"][str(u)]["IPos"]
            neg, seen = b["INeg"][r].tolist(), set(ds.train_seq(u).tolist())
            assert all((n == 0 and p == 0) or n not in seen for n, p in zip(neg, b["ISeq"][r].tolist()))
    assert sorted(got) == sorted(fx["train"])            # user 2's train sequence has one item: no (input, target) pair, no row
    for mode in ("valid", "test"):
        for b in EvalSampler(ds, S, 8, mode):
            for r, u in enumerate(b["User"].tolist()):
                want = fx[mode][str(u)]
                assert b["ISeq"][r].tolist() == want["ISeq"] and b["IUnseen"][r] == want["IUnseen"]
                assert sorted(set(b["ISeen"][r])) == want["ISeen"]


def test_scripts_parse_and_cited_scripts_exist():
    """scripts/ holds the measurement tooling DESIGN.md / README cite: every python file parses, every cited file exists."""
    import ast, glob, os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in glob.glob(os.path.join(root, "scripts", "*.py")):
        ast.parse(open(f).read(), f)
    cited = set()
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        cited |= set(re.findall(r"scripts/([A-Za-z0-9_]+\.(?:py|sh))", open(os.path.join(root, doc)).read()))
    missing = sorted(c for c in cited if not os.path.exists(os.path.join(root, "scripts", c)))
    assert not missing, missing


def test_oracle_device_sampler_restatement_gives_the_hand_derived_rows():
    """oracle/sampler.py (the numpy restatement the GPU sampler is checked against bit for bit) on tests/golden/sampler_rows.json."""
    import json
    import os
    import numpy as np
    from oracle import sampler as osm
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampler_rows.json")))
    train = [np.asarray(s[:-2], np.int64) for s in fx["seqs"]]
    ptr = np.zeros(len(train) + 1, np.int64)
    np.cumsum([len(s) for s in train], out=ptr[1:])
    users, seq, pos, neg = osm.seq_train_sample(ptr, np.concatenate(train), np.arange(len(train)), 0, len(train), fx["maxlen"], fx["num_items"], 5, 3)
    for u, want in fx["train"].items():
        assert seq[int(u)].tolist() == want["ISeq"] and pos[int(u)].tolist() == want["IPos"]
        live = seq[int(u)] > 0
        assert not np.isin(neg[int(u)][live], train[int(u)]).any() and (neg[int(u)][~live] == 0).all()
