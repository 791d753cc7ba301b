"""GPU tests of score_topk's split form (bf16 hi/mid screening on the XDL pipe + exact re-scoring + certificate + exact
fallback; csrc/score.hip): it must return exactly what the exact fp32-MFMA kernel and the C oracle return, its error
bound must hold with room to spare, and the cases the certificate cannot decide (ties at the K-th score) must take the
fallback and still be right.  Reference contract: Coach.evaluate, mirrored at UniSRec/main.py:408-414."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import os
    from recboard_amd import lib, ops
    # the A/B switches exist only in the diagnostic twin of the library (make dbg: -DRE_DEBUG); the product .so has none.
    # The same kernels, the same entry points: this module runs `ops` on the twin and puts the product library back afterwards.
    prod_path, prod_lib = lib.LIB_PATH, lib._LIB
    lib.LIB_PATH, lib._LIB = os.path.join(os.path.dirname(lib.LIB_PATH), "librecengine_dbg.so"), None
    L = lib.load()
    L.re_dbg_score_x2.argtypes = [ctypes.c_int]; L.re_dbg_score_x2.restype = None
    L.re_dbg_score_x2_maxerr.argtypes = [ctypes.c_int]; L.re_dbg_score_x2_maxerr.restype = None
    L.re_dbg_score_x2_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.re_dbg_score_x2_stats.restype = None
    L.re_dbg_score_sample.argtypes = [ctypes.c_int]; L.re_dbg_score_sample.restype = None
    yield ops, L
    L.re_dbg_score_x2(1); L.re_dbg_score_x2_maxerr(0); L.re_dbg_score_sample(1)
    lib.LIB_PATH, lib._LIB = prod_path, prod_lib


def stats(L):
    """-> (users sent to the exact fallback, max |s' - s| / eps) since the last call; resets the counters."""
    torch.cuda.synchronize()
    out = (ctypes.c_uint32 * 2)()
    L.re_dbg_score_x2_stats(out, 1)
    return int(out[0]), float(np.array([out[1]], np.uint32).view(np.float32)[0])


def seen_csr(g, U, N, n):
    sp = torch.arange(0, U + 1, device="cuda") * n
    si = torch.sort(torch.randint(0, N, (U, n), device="cuda", generator=g), 1).values.reshape(-1)
    return sp, si


def both_paths(ops, L, q, E, sp, si, K, prep=None):
    L.re_dbg_score_x2(0)
    v0, i0 = ops.score_topk(q, E, sp, si, K)
    L.re_dbg_score_x2(1)
    stats(L)
    L.re_dbg_score_x2_maxerr(1)
    v1, i1 = ops.score_topk(q, E, sp, si, K, prep=prep)
    flagged, ratio = stats(L)
    L.re_dbg_score_x2_maxerr(0)
    return (v0, i0), (v1, i1), flagged, ratio


@pytest.mark.parametrize("K", [50, 10, 26, 27, 1])
def test_split_equals_exact_kernel_full_beauty(env, K):
    """All 22 363 x 12 101 scores of BASELINE configs[1]: the split form and the exact kernel agree bit for bit (values and
    indices), nobody needs the fallback on generic scores, and the observed screening error stays far inside the bound the
    certificate uses (max |s' - s| / eps < 0.5; measured 0.08)."""
    ops, L = env
    g = torch.Generator(device="cuda").manual_seed(K)
    U, N, D = 22363, 12101, 64
    q = torch.randn(U, D, device="cuda", generator=g)
    E = torch.randn(N, D, device="cuda", generator=g)
    sp, si = seen_csr(g, U, N, 8)
    (v0, i0), (v1, i1), flagged, ratio = both_paths(ops, L, q, E, sp, si, K)
    assert torch.equal(i0, i1)
    assert torch.equal(v0.view(torch.int32), v1.view(torch.int32))
    assert flagged == 0
    assert 0.0 < ratio < 0.5


def test_split_equals_exact_kernel_d128(env):
    """D = 128 (BASELINE configs[4]: one workgroup per CU, two 32 KB stage buffers): a 4 096-user batch against 50 000 items and a
    512-user batch against a long shard -- the split form and the exact kernel agree bit for bit."""
    ops, L = env
    g = torch.Generator(device="cuda").manual_seed(128)
    for U, N, n_seen in ((4096, 50000, 16), (512, 1_500_000, 0)):
        q = torch.randn(U, 128, device="cuda", generator=g)
        E = torch.randn(N, 128, device="cuda", generator=g)
        sp, si = seen_csr(g, U, N, n_seen) if n_seen else (None, None)
        (v0, i0), (v1, i1), flagged, ratio = both_paths(ops, L, q, E, sp, si, 50)
        assert torch.equal(i0, i1) and torch.equal(v0.view(torch.int32), v1.view(torch.int32))
        assert flagged == 0
        assert 0.0 < ratio < 0.5


def test_split_equals_exact_kernel_long_catalog(env):
    """Few users against a long catalog (a config-5 shard): the sliced work split, three stage buffers with requests two stages
    ahead, the sample bound computed in chunks -- the split form and the exact kernel agree bit for bit, with and without a
    seen mask."""
    ops, L = env
    g = torch.Generator(device="cuda").manual_seed(64)
    U, N = 512, 2_500_000
    q = torch.randn(U, 64, device="cuda", generator=g)
    E = torch.randn(N, 64, device="cuda", generator=g)
    for sp, si in (seen_csr(g, U, N, 40), (None, None)):
        (v0, i0), (v1, i1), flagged, ratio = both_paths(ops, L, q, E, sp, si, 50)
        assert torch.equal(i0, i1) and torch.equal(v0.view(torch.int32), v1.view(torch.int32))
        assert flagged == 0 and 0.0 < ratio < 0.5


def test_split_on_trained_like_state_with_popular_head(env):
    """Scores dominated by a run of neighbouring ids (popular items sit at the low ids in the bench's synthetic data and in
    many real catalogs): every user's best 50 come from the same 64 items, i.e. one stage of the kernel.  The lists must
    take that without sending anybody to the fallback."""
    ops, L = env
    g = torch.Generator(device="cuda").manual_seed(3)
    U, N, D, K = 8192, 12101, 64, 50
    E = 0.05 * torch.randn(N, D, device="cuda", generator=g)
    pop = torch.randn(D, device="cuda", generator=g)
    E[:64] += 0.5 * pop * torch.linspace(1.5, 1.0, 64, device="cuda")[:, None]      # a popular head of 64 consecutive ids
    q = pop[None, :] + 0.3 * torch.randn(U, D, device="cuda", generator=g)
    sp, si = seen_csr(g, U, N, 6)
    (v0, i0), (v1, i1), flagged, ratio = both_paths(ops, L, q, E, sp, si, K)
    assert torch.equal(i0, i1) and torch.equal(v0, v1)
    assert int((i1 < 64).sum()) > 0.9 * U * K          # the head really is what the lists hold
    assert flagged == 0
    assert ratio < 0.5


def test_split_falls_back_on_ties_and_stays_exact(env):
    """Every row 8 times in the catalog (ties around the K-th score) and users whose scores are all exactly 0: where the
    certificate cannot separate the K-th score from what was dropped, the users go to the exact kernel -- and the result is
    the oracle's (ties -> lowest index)."""
    from oracle import ranking
    ops, L = env
    rng = np.random.default_rng(5)
    U, N, D, K = 2304, 1536, 64, 50
    base = rng.standard_normal((N // 8, D)).astype(np.float32)
    E = np.ascontiguousarray(np.tile(base, (8, 1))[rng.permutation(N)])
    Q = rng.standard_normal((U, D)).astype(np.float32)
    Q[::11] = 0.0                                          # and some users for whom every score is exactly 0
    d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    L.re_dbg_score_x2(1)
    stats(L)
    v, i = ops.score_topk(d(Q), d(E), None, None, K)
    flagged, _ = stats(L)
    rv, ri = ranking.score_topk(Q, E, None, None, K)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(v.cpu().numpy(), rv)
    assert flagged >= len(range(0, U, 11))                 # at least the all-zero users


def test_prepared_table_and_sampled_thresholds(env):
    """re_score_prepare + re_score_topk_prepared (the table split once, many user batches) return what re_score_topk returns;
    so does the split form without its sampled starting thresholds (they only change how much work the lists do)."""
    ops, L = env
    g = torch.Generator(device="cuda").manual_seed(9)
    U, N, D, K = 4096, 30011, 64, 50
    q = torch.randn(U, D, device="cuda", generator=g)
    E = torch.randn(N, D, device="cuda", generator=g)
    sp, si = seen_csr(g, U, N, 12)
    L.re_dbg_score_x2(1)
    v, i = ops.score_topk(q, E, sp, si, K)
    prep = ops.score_prepare(E)
    assert prep is not None
    for lo in (0, 2048):                                   # two user batches against one prepared table
        sl = slice(lo, lo + 2048)
        spb = sp[lo:lo + 2049] - sp[lo]
        sib = si[int(sp[lo]):int(sp[lo + 2048])]
        vb, ib = ops.score_topk(q[sl].contiguous(), E, spb.contiguous(), sib.contiguous(), K, prep=prep)
        assert torch.equal(ib, i[sl]) and torch.equal(vb, v[sl])
    L.re_dbg_score_sample(0)
    v2, i2 = ops.score_topk(q, E, sp, si, K)
    L.re_dbg_score_sample(1)
    assert torch.equal(i2, i) and torch.equal(v2, v)
    assert ops.score_prepare(torch.randn(100, 48, device="cuda")) is None      # no split form for this D: exact path


def test_split_adversarial_scales(env):
    """Inputs that stretch the error bound: tables of magnitude 1e-4 (MF-BPR's init, MF-BPR/main.py:55), components that
    differ by 1e4 inside one row (cancellation), half-integer values (mass ties).  Results equal the exact kernel's; the
    observed error stays inside the bound."""
    ops, L = env
    g = torch.Generator(device="cuda").manual_seed(11)
    U, N, D, K = 4096, 9000, 64, 50
    q = torch.randn(U, D, device="cuda", generator=g)
    E = torch.randn(N, D, device="cuda", generator=g)
    sp, si = seen_csr(g, U, N, 8)
    Ec = E.clone(); Ec[:, :32] *= 100.0
    qc = q.clone(); qc[:, 32:] *= 100.0
    cases = [(q * 1e-4, E * 1e-4), (qc, Ec), (torch.round(q * 2) / 2, torch.round(E * 2) / 2)]
    for qq, EE in cases:
        (v0, i0), (v1, i1), flagged, ratio = both_paths(ops, L, qq.contiguous(), EE.contiguous(), sp, si, K)
        assert torch.equal(i0, i1) and torch.equal(v0, v1)
        assert ratio < 1.0


@pytest.mark.parametrize("D,N", [(64, 12101), (128, 5000)])
def test_front_launch_gives_the_three_launches_results_and_few_fallbacks(env, D, N):
    """score_front_k (query split + item split + starting thresholds in ONE launch, round 5) against the three launches it replaces
    (re_dbg_score_front(0)): identical values and indices -- the planes it writes are score_split_k's bit for bit, its thresholds are only filter
    values -- and its cheaper sample arithmetic (hi planes, pair maxima) must not send more than a handful of users to the exact fallback.
    Enough users for the sample not to be cut into chunks (> 16 384): the shape the front launch takes."""
    ops, L = env
    L.re_dbg_score_front.argtypes = [ctypes.c_int]; L.re_dbg_score_front.restype = None
    g = torch.Generator(device="cuda").manual_seed(21)
    U, K = 17000, 50
    q = torch.randn(U, D, device="cuda", generator=g)
    E = torch.randn(N, D, device="cuda", generator=g)
    sp, si = seen_csr(g, U, N, 6)
    try:
        L.re_dbg_score_front(0)
        stats(L)
        v0, i0 = ops.score_topk(q, E, sp, si, K)
        f0, _ = stats(L)
        L.re_dbg_score_front(1)
        v1, i1 = ops.score_topk(q, E, sp, si, K)
        f1, _ = stats(L)
    finally:
        L.re_dbg_score_front(1)
    assert torch.equal(i0, i1) and torch.equal(v0, v1)
    assert f1 <= f0 + 8, (f0, f1)          # (flagged users cost time, never correctness)
    # and against the exact kernel
    L.re_dbg_score_x2(0)
    ve, ie = ops.score_topk(q, E, sp, si, K)
    L.re_dbg_score_x2(1)
    assert torch.equal(ie, i1) and torch.equal(ve, v1)
