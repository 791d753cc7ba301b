"""GPU: re_sasrec_batch_prep -- the per-batch preparation of a SASRec step as one launch (SASRec/main.py:199-204: mask of the
non-pad positions, their number, the item rows the step touches) and the encoder's work plan (csrc/enc_common.h)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from recboard_amd import ops as _ops
    return _ops


def _batch(seed, B, S, N, kind):
    rng = np.random.default_rng(seed)
    if kind == "beauty":
        lens = np.clip(rng.geometric(1 / 5.9, B) + 1, 1, S - 1)
    elif kind == "uniform":
        lens = rng.integers(0, S + 1, B)          # empty and full sequences included
    else:
        lens = np.full(B, S)
    seq = np.zeros((B, S), np.int64)
    for b in range(B):
        if lens[b]:
            seq[b, S - lens[b]:] = rng.integers(1, N + 1, lens[b])
    if kind == "uniform" and B > 3:
        seq[3, S - 5] = 0                           # a pad INSIDE a sequence: an explicit row, still a key
    pos = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    return seq, pos, neg


def _plan_arrays(pb, B, S):
    w = pb.plan.view(torch.int32).cpu().numpy()
    mt = B * ((S + 15) // 16)
    rm0 = (8 + mt + 1) // 2 * 2
    hdr, items = w[:8], w[8:8 + mt]
    rowmap = w[rm0:rm0 + 2 * 16 * mt].reshape(-1, 2)
    return hdr, items, rowmap


@pytest.mark.parametrize("B,S,kind", [(512, 50, "beauty"), (37, 50, "uniform"), (1500, 50, "uniform"), (9, 64, "full"), (1, 7, "uniform")])
def test_batch_prep_matches_the_reference_expressions_and_plan_is_a_partition(ops, B, S, kind):
    N = 300
    seq, pos, neg = _batch(B + S, B, S, N, kind)
    d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    pb = ops.sasrec_batch_prep(d(seq), d(pos), d(neg))
    v = seq.reshape(-1) != 0
    np.testing.assert_array_equal(pb.valid.cpu().numpy(), v.astype(np.uint8))
    assert int(pb.count) == int(v.sum())
    np.testing.assert_array_equal(pb.rows_all.cpu().numpy(),
                                  np.concatenate([seq.reshape(-1), np.where(v, pos.reshape(-1) + 1, 0), np.where(v, neg.reshape(-1) + 1, 0)]))
    hdr, items, rowmap = _plan_arrays(pb, B, S)
    n_items, n_tiles, n_long, G = (int(x) for x in hdr[:4])
    assert int(hdr[4]) == int(v.sum())
    ncu = ops.num_cus()
    # items: consecutive tile ranges that cover [0, n_tiles) exactly once, long ones (one sequence each) first and largest first
    t = 0
    sizes = []
    for i in range(n_items):
        tile0, nt, kind_ = int(items[i]) & 0xFFFFFF, (int(items[i]) >> 24) & 0xF, (int(items[i]) >> 28) & 0xF
        assert tile0 == t and 1 <= nt <= 4 and kind_ == (1 if i < n_long else 0)
        if i >= n_long and i < n_items - 1:
            assert nt == G
        t += nt
        sizes.append(nt)
    assert t == n_tiles
    assert sizes[:n_long] == sorted(sizes[:n_long], reverse=True)
    if n_tiles - sum(sizes[:n_long]) > max(ncu - n_long, 1):
        assert G > 1 or n_items <= ncu
    # rows: every position from a sequence's first real token on appears exactly once (an empty sequence: its last position)
    first = np.array([(np.nonzero(seq[b])[0][0] if seq[b].any() else S - 1) for b in range(B)])
    want = {(b * S + s) for b in range(B) for s in range(first[b], S)}
    got = rowmap[:16 * n_tiles]
    real = got[got[:, 0] >= 0]
    assert len(real) == len(want) and set(real[:, 0].tolist()) == want
    np.testing.assert_array_equal(real[:, 1], first[real[:, 0] // S])
    # a sequence's rows are consecutive and ascending; short sequences stay inside one tile, a long one starts at its item's tile
    rows_of = {}
    for r, g in enumerate(got[:, 0]):
        if g >= 0:
            rows_of.setdefault(int(g) // S, []).append((r, int(g)))
    for b, lst in rows_of.items():
        rr = [r for r, _ in lst]
        assert rr == list(range(rr[0], rr[0] + len(rr))) and [g for _, g in lst] == list(range(b * S + first[b], (b + 1) * S))
        if len(rr) <= 16:
            assert rr[0] // 16 == rr[-1] // 16
        else:
            assert rr[0] % 16 == 0
    # packing: complement pairing leaves few dummy rows among the short sequences -- within 4 % (+ one tile) of a perfect packing
    spans = np.array([S - first[b] for b in range(B)])
    short_rows, long_tiles = int(spans[spans <= 16].sum()), int(sum(-(-int(x) // 16) for x in spans[spans > 16]))
    assert n_tiles <= long_tiles + int(np.ceil(short_rows * 1.04 / 16)) + 1, (n_tiles, long_tiles, short_rows)
    # the same launch, staging into a static blob with the step scalars (the captured step's staging launch)
    blob = torch.zeros(ops.prep_layout(B, S)[1], dtype=torch.uint8, device="cuda")
    state = torch.zeros(4, dtype=torch.int32, device="cuda")
    pb2 = ops.sasrec_batch_prep(d(seq), d(pos), d(neg), blob=blob, state=state, seed=1234, step=3, lr=5e-4)
    for a, b_ in ((pb2.seq, seq), (pb2.pos, pos), (pb2.neg, neg)):
        np.testing.assert_array_equal(a.cpu().numpy(), b_)
    assert torch.equal(pb2.rows_all, pb.rows_all) and torch.equal(pb2.plan[:4 * (8 + n_items)], pb.plan[:4 * (8 + n_items)])
    st = state.cpu().numpy()
    assert st[0] == 1234 and st[1] == 0
    hy = st[2:].view(np.float32)
    np.testing.assert_allclose(hy, [5e-4 / (1 - 0.9 ** 3), 1 / np.sqrt(1 - 0.999 ** 3)], rtol=1e-6)
