"""GPU tests of the autograd surface (recboard_amd/nn.py): the engine's operators as differentiable torch ops must give what
the reference's aten ops give -- values and gradients -- on the same inputs (tolerance 1e-4 relative, the north star's)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rnn():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from recboard_amd import nn as rnn_
    return rnn_


def close(a, b, tol=1e-4):
    a, b = a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy()
    return np.abs(a - b).max() <= tol * (np.abs(b).max() + 1e-12)


def test_embedding_matches_torch_embedding(rnn):
    """nn.Embedding.__call__ + embedding_dense_backward (MF-BPR/main.py:84-86; SASRec/main.py:183): values bit-equal,
    dense table gradient equal incl. heavy duplicates, padding row without gradient."""
    g = torch.Generator(device="cuda").manual_seed(0)
    R, D = 3001, 64
    ref = torch.nn.Embedding(R, D, padding_idx=0).cuda()
    mine = rnn.Embedding(R, D, padding_idx=0, device="cuda")
    with torch.no_grad():
        mine.weight.copy_(ref.weight)
    idx = torch.randint(0, R, (257, 50), device="cuda", generator=g)
    idx[:, :20] = torch.randint(0, 8, (257, 20), device="cuda", generator=g)      # hot rows + the padding row
    w = torch.randn(257, 50, D, device="cuda", generator=g)
    ya, yb = ref(idx), mine(idx)
    assert torch.equal(ya, yb)
    (ya * w).sum().backward(); (yb * w).sum().backward()
    assert close(mine.weight.grad, ref.weight.grad, 1e-5)
    assert float(mine.weight.grad[0].abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rnn.gather_rows(torch.zeros(4, 8), torch.zeros(3, dtype=torch.long))


def test_bpr_triplet_matches_reference_formula(rnn):
    """MF.fit (MF-BPR/main.py:81-93): gather u, i+, i-; two row dots; BPRLoss = mean softplus(neg - pos)."""
    g = torch.Generator(device="cuda").manual_seed(1)
    U, N, D, n = 500, 700, 64, 2048
    Ut = (0.1 * torch.randn(U, D, device="cuda", generator=g)).requires_grad_()
    It = (0.1 * torch.randn(N, D, device="cuda", generator=g)).requires_grad_()
    users = torch.randint(0, U, (n, 1), device="cuda", generator=g)
    pos = torch.randint(0, 40, (n, 1), device="cuda", generator=g)              # popular positives: many collisions
    neg = torch.randint(0, N, (n, 1), device="cuda", generator=g)
    loss = rnn.bpr_triplet(Ut, It, users, pos, neg)
    loss.backward()
    gU, gI = Ut.grad.clone(), It.grad.clone()
    Ut.grad = It.grad = None
    u, ip, ineg = Ut[users.view(-1)], It[pos.view(-1)], It[neg.view(-1)]
    ref = torch.nn.functional.softplus((u * ineg).sum(-1) - (u * ip).sum(-1)).mean()
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    assert close(gU, Ut.grad) and close(gI, It.grad)
    crit = rnn.BPRLoss()
    assert abs(float(crit((u * ip).sum(-1), (u * ineg).sum(-1))) - float(ref)) <= 1e-6


def test_score_full_values_and_gradients(rnn):
    """recommend_from_full (SASRec/main.py:223-228) and its use under a CE over the catalog (SASRec/main.py:217-219)."""
    g = torch.Generator(device="cuda").manual_seed(2)
    B, N, D = 300, 1201, 64
    Q = torch.randn(B, D, device="cuda", generator=g, requires_grad=True)
    E = torch.randn(N, D, device="cuda", generator=g, requires_grad=True)
    y = torch.randint(0, N, (B,), device="cuda", generator=g)
    S = rnn.score_full(Q, E)
    torch.nn.functional.cross_entropy(S, y).backward()
    gQ, gE = Q.grad.clone(), E.grad.clone()
    Q.grad = E.grad = None
    Sr = Q @ E.t()
    torch.nn.functional.cross_entropy(Sr, y).backward()
    assert close(S, Sr, 1e-5) and close(gQ, Q.grad) and close(gE, E.grad)


def test_spmm_sym_matches_torch_sparse(rnn):
    """Adj @ X with the symmetric normalised adjacency (LightGCN/main.py:77-86): forward and the gradient A^T g = A g."""
    rng = np.random.default_rng(3)
    n, D, nnz = 900, 64, 6000
    r, c = rng.integers(0, n, nnz), rng.integers(0, n, nnz)
    A = np.zeros((n, n), np.float32)
    A[r, c] = rng.random(nnz).astype(np.float32)
    A = A + A.T
    At = torch.from_numpy(A).cuda()
    csr = At.to_sparse_csr()
    crow, col, val = csr.crow_indices().contiguous(), csr.col_indices().contiguous(), csr.values().contiguous()
    X = torch.randn(n, D, device="cuda", requires_grad=True)
    w = torch.randn(n, D, device="cuda")
    Y = rnn.spmm_sym(crow, col, val, X)
    (Y * w).sum().backward()
    gX = X.grad.clone()
    X.grad = None
    Yr = At @ X
    (Yr * w).sum().backward()
    assert close(Y, Yr) and close(gX, X.grad)


def test_mf_bpr_trains_like_the_aten_model(rnn):
    """A FreeRec-style MF module (MF-BPR/main.py:36-93) written with torch.nn.Embedding and with the engine's Embedding,
    three Adam steps each: same losses, same tables."""
    U, N, D, n = 400, 600, 64, 1024
    g = torch.Generator(device="cuda").manual_seed(4)

    class MF(torch.nn.Module):
        def __init__(self, emb):
            super().__init__()
            self.User, self.Item = emb(U, D), emb(N, D)
            self.criterion = rnn.BPRLoss()

        def forward(self, users, pos, neg):
            u, ip, ineg = self.User(users), self.Item(pos), self.Item(neg)
            return self.criterion(torch.einsum("BKD,BKD->BK", u, ip), torch.einsum("BKD,BKD->BK", u, ineg))

    a = MF(lambda r, d: torch.nn.Embedding(r, d)).cuda()
    b = MF(lambda r, d: rnn.Embedding(r, d, device="cuda"))
    with torch.no_grad():
        for pa, pb in zip(a.parameters(), b.parameters()):
            pa.mul_(0.01); pb.copy_(pa)
    oa, ob = torch.optim.Adam(a.parameters(), lr=1e-2), torch.optim.Adam(b.parameters(), lr=1e-2)
    for _ in range(3):
        users = torch.randint(0, U, (n, 1), device="cuda", generator=g)
        pos = torch.randint(0, N, (n, 1), device="cuda", generator=g)
        neg = torch.randint(0, N, (n, 1), device="cuda", generator=g)
        la, lb = a(users, pos, neg), b(users, pos, neg)
        oa.zero_grad(); ob.zero_grad()
        la.backward(); lb.backward()
        oa.step(); ob.step()
        assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(la))
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert close(pb, pa)


def test_registered_custom_ops_match_aten_and_pass_opcheck(rnn):
    """torch.ops.recengine.* called directly (SURVEY.md §8b torch-op surface): values and gradients against the aten ops the reference
    dispatches to, and torch.library.opcheck (schema, fake implementation, autograd registration) on each differentiable op."""
    R = torch.ops.recengine
    g = torch.Generator(device="cuda").manual_seed(11)
    W = torch.randn(300, 64, device="cuda", generator=g, requires_grad=True)
    idx = torch.randint(0, 300, (40, 7), device="cuda", generator=g)
    y = R.gather_rows(W, idx)
    assert torch.equal(y, W[idx])
    w = torch.randn_like(y)
    (gW,) = torch.autograd.grad((y * w).sum(), W)
    (rW,) = torch.autograd.grad((W[idx] * w).sum(), W)
    assert close(gW, rW, 1e-5)
    # scatter_add_rows is differentiable too (its gradient is a gather)
    gsrc = torch.randn(280, 64, device="cuda", generator=g, requires_grad=True)
    didx = torch.randint(0, 300, (280,), device="cuda", generator=g)
    dW = R.scatter_add_rows(gsrc, didx, 300, -1)
    ref = torch.zeros(300, 64, device="cuda").index_add_(0, didx, gsrc.detach())
    assert close(dW, ref, 1e-5)
    (gg,) = torch.autograd.grad((dW * W.detach()).sum(), gsrc)
    assert close(gg, W.detach()[didx], 1e-6)
    Q = torch.randn(33, 64, device="cuda", generator=g, requires_grad=True)
    E = torch.randn(500, 64, device="cuda", generator=g, requires_grad=True)
    S = R.score_dense(Q, E)
    assert close(S, Q @ E.T, 1e-5)
    gq, ge = torch.autograd.grad(S.logsumexp(1).sum(), (Q, E))
    rq, re_ = torch.autograd.grad((Q @ E.T).logsumexp(1).sum(), (Q, E))
    assert close(gq, rq) and close(ge, re_)
    sp = torch.arange(0, 34, device="cuda") * 2
    si = torch.sort(torch.randint(0, 500, (33, 2), device="cuda", generator=g), 1).values.reshape(-1)
    v, i = R.score_topk(Q.detach(), E.detach(), sp, si, 20)
    sc = (Q @ E.T).detach()
    sc[torch.arange(33, device="cuda").repeat_interleave(2), si] = -1e23
    rv, ri = torch.topk(sc, 20, dim=1)
    assert close(v, rv, 1e-5) and float((i == ri).float().mean()) > 0.99
    users, pos, neg = (torch.randint(0, 300, (64,), device="cuda", generator=g) for _ in range(3))
    for op, args in ((R.gather_rows, (W, idx)), (R.score_dense, (Q, E)), (R.bpr_triplet, (W, W.detach().clone().requires_grad_(), users, pos, neg))):
        torch.library.opcheck(op, args, test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))


def test_two_streams_call_the_library_concurrently(rnn):
    """The C ABI is re-entrant (no global mutable state): the same entry points enqueued on two streams at once give the results of
    serial calls."""
    from recboard_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    Q = [torch.randn(2048, 64, device="cuda", generator=g) for _ in range(2)]
    E = [torch.randn(9000 + 500 * k, 64, device="cuda", generator=g) for k in range(2)]
    W = torch.randn(50000, 64, device="cuda", generator=g)
    idx = [torch.randint(0, 50000, (200000,), device="cuda", generator=g) for _ in range(2)]
    serial = [(ops.score_topk(Q[k], E[k], None, None, 50), ops.gather_rows(W, idx[k])) for k in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    out = [None, None]
    for rep in range(3):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                out[k] = (ops.score_topk(Q[k], E[k], None, None, 50), ops.gather_rows(W, idx[k]))
        torch.cuda.synchronize()
        for k in range(2):
            assert torch.equal(out[k][0][1], serial[k][0][1]) and torch.equal(out[k][0][0], serial[k][0][0])
            assert torch.equal(out[k][1], serial[k][1])
