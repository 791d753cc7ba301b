"""GPU property tests: gather/scatter-add with duplicates, padding and empty segments; ragged and
degenerate encoder inputs; top-K with K > #unmasked."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from recboard_amd import ops as _ops
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(R=st.integers(1, 300), n=st.integers(0, 700), D=st.sampled_from([1, 4, 10, 64, 128]), seed=st.integers(0, 10 ** 6),
       pad=st.sampled_from([-1, 0]))
def test_scatter_is_the_adjoint_of_gather(ops, R, n, D, seed, pad):
    """<gather(W, idx), G> == <W, scatter_add(G, idx)> for every W, G (padding rows excluded); bit-exact gather."""
    rng = np.random.default_rng(seed)
    W = rng.standard_normal((R, D)).astype(np.float32)
    idx = rng.integers(0, R, n)
    G = rng.standard_normal((n, D)).astype(np.float32)
    out = ops.gather_rows(dev(W), dev(idx)).cpu().numpy()
    np.testing.assert_array_equal(out, W[idx])
    dW = ops.scatter_add_rows(dev(G), dev(idx), R, pad).cpu().numpy()
    keep = idx != pad
    lhs = float((out[keep].astype(np.float64) * G[keep]).sum())
    rhs = float((W.astype(np.float64) * dW).sum())
    assert abs(lhs - rhs) <= 1e-4 * (abs(lhs) + 1.0)
    if pad >= 0:
        assert (dW[pad] == 0).all()


def test_encoder_degenerate_batches(ops):
    """all-pad sequences, a single sequence, B not a multiple of 4 (partial short work item), dropout on."""
    from oracle import sasrec as osas
    from recboard_amd.sasrec import param_shapes
    g = torch.Generator().manual_seed(0)
    P = {k: (torch.randn(s, generator=g) * 0.1 + (1.0 if ("LN" in k and k.endswith("weight")) else 0.0)) for k, s in param_shapes(50, 50, 64, 2).items()}
    Pd = {k: v.cuda() for k, v in P.items()}
    for B in (1, 3, 5):
        seq = torch.zeros(B, 50, dtype=torch.long)
        if B > 1:
            seq[1, -4:] = torch.tensor([3, 9, 1, 7])
        if B > 3:
            seq[4, :] = 5
        x0 = ops.sasrec_embed(Pd["Item.embeddings.weight"], Pd["Position.weight"], seq.cuda(), 8.0)
        u, _ = ops.sasrec_encoder_fwd(x0, seq.cuda(), ops.sasrec_block_tensors(Pd, 2), Pd["lastLN.weight"], Pd["lastLN.bias"], 2)
        with torch.no_grad():
            ref, _ = osas.encode(P, seq, 2)
        torch.testing.assert_close(u.cpu(), ref, rtol=1e-4, atol=2e-5)
        assert torch.isfinite(u).all()


def test_topk_all_items_masked_and_tiny_catalog(ops):
    from oracle import ranking
    Q = np.random.default_rng(1).standard_normal((3, 64)).astype(np.float32)
    E = np.random.default_rng(2).standard_normal((5, 64)).astype(np.float32)
    sp, si = np.array([0, 5, 5, 7]), np.array([0, 1, 2, 3, 4, 1, 3])
    v, i = ops.score_topk(dev(Q), dev(E), dev(sp), dev(si), 8)            # K > N and user 0 has everything masked
    rv, ri = ranking.score_topk(Q, E, sp, si, 8)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(v.cpu().numpy(), rv)
