"""GPU: the fused kernels as torch custom ops a model file can call directly -- `recengine::sasrec_encoder` (SASRec.encode: SASRec/main.py:163-193),
`recengine::bce_pair` (the pair criteria: :205-215), `recengine::fm_bag` (DeepFM's front end: DeepFM/main.py:58-62,80-85,204-206): values
and gradients against the reference golden / a plain torch restatement, and torch.library.opcheck on each."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sasrec_params(z, L):
    from recboard_amd import ops
    named = {k[6:]: torch.from_numpy(z[k]).cuda().requires_grad_(True) for k in z.files if k.startswith("param/") and z[k].dtype == np.float32}
    blocks = ops.sasrec_block_tensors(named, L)
    return named, list(blocks) + [named["lastLN.weight"], named["lastLN.bias"]]


def test_sasrec_encoder_and_bce_pair_ops_reproduce_the_golden_step():
    import recboard_amd.torch_ops  # noqa: F401
    R = torch.ops.recengine
    z = np.load(os.path.join(G, "sasrec_bce.npz"))
    L, D = int(z["cfg/num_blocks"]), int(z["cfg/D"])
    named, params = _sasrec_params(z, L)
    E, P = named["Item.embeddings.weight"], named["Position.weight"]
    seq, pos, neg = (torch.from_numpy(z[k]).cuda() for k in ("in/seq", "in/pos", "in/neg"))
    u = R.sasrec_encoder(E, P, seq, params, float(D ** 0.5), 0.0, 0)[0]
    keep = seq != 0
    np.testing.assert_allclose(u[keep].detach().cpu().numpy(), z["out/userEmbds"][keep.cpu().numpy()], rtol=1e-4, atol=2e-5)
    loss = R.bce_pair(u.reshape(-1, D), E, pos, neg, keep, 0, 1)[0]
    assert abs(loss.item() - float(z["out/rec_loss"])) <= 1e-5 * abs(float(z["out/rec_loss"]))
    loss.backward()
    for k, p in named.items():
        if p.grad is None:
            continue
        ref = z["grad/" + k].reshape(p.shape)
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err <= 1e-4 * np.abs(ref).max() + 1e-7, (k, err)
    assert (E.grad[0] == 0).all()


def _fm_bag_torch(T, TL, b, offsets, x):
    rows = x + offsets.unsqueeze(0)
    e = T[rows]                                            # [B, F, D]
    fm = 0.5 * (e.sum(1).pow(2) - e.pow(2).sum(1)).sum(-1)
    return e.reshape(x.shape[0], -1), TL[rows].sum(1) + b + fm


def test_fm_bag_op_matches_a_torch_restatement_with_gradients():
    import recboard_amd.torch_ops  # noqa: F401
    g = torch.Generator(device="cuda").manual_seed(3)
    counts = torch.tensor([957, 4082, 7, 7, 2, 3, 2, 9, 80, 233])
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)[:-1]]).cuda()
    Rr, D, B = int(counts.sum()), 10, 512
    x = torch.stack([torch.randint(0, int(c), (B,), device="cuda", generator=g) for c in counts], 1)
    T = (torch.randn(Rr, D, device="cuda", generator=g) * 0.1).requires_grad_(True)
    TL = (torch.randn(Rr, device="cuda", generator=g) * 0.1).requires_grad_(True)
    b = torch.zeros(1, device="cuda").requires_grad_(True)
    E, fl = torch.ops.recengine.fm_bag(T, TL, b, offsets, x)
    w1, w2 = torch.randn_like(E), torch.randn_like(fl)
    (E * w1).sum().add((fl * w2).sum()).backward()
    got = [t.grad.clone() for t in (T, TL, b)]
    for t in (T, TL, b):
        t.grad = None
    Er, fr = _fm_bag_torch(T, TL, b, offsets, x)
    (Er * w1).sum().add((fr * w2).sum()).backward()
    torch.testing.assert_close(E, Er, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(fl, fr, rtol=1e-5, atol=1e-6)
    for a, t in zip(got, (T, TL, b)):
        torch.testing.assert_close(a, t.grad, rtol=1e-4, atol=1e-5)


def test_bce_pair_op_matches_torch_and_every_new_op_passes_opcheck():
    import recboard_amd.torch_ops  # noqa: F401
    R = torch.ops.recengine
    g = torch.Generator(device="cuda").manual_seed(5)
    n, D, N = 300, 64, 90
    U = torch.randn(n, D, device="cuda", generator=g).requires_grad_(True)
    E = (torch.randn(N + 1, D, device="cuda", generator=g) * 0.3).requires_grad_(True)
    pos, neg = torch.randint(0, N, (n,), device="cuda", generator=g), torch.randint(0, N, (n,), device="cuda", generator=g)
    valid = torch.rand(n, device="cuda", generator=g) < 0.7
    for kind in (0, 1):
        loss = R.bce_pair(U, E, pos, neg, valid, kind, 1)[0]
        loss.backward()
        gu, ge = U.grad.clone(), E.grad.clone()
        U.grad = E.grad = None
        pl, nl = (U[valid] * E[1 + pos[valid]]).sum(-1), (U[valid] * E[1 + neg[valid]]).sum(-1)
        F = torch.nn.functional
        if kind == 0:
            ref = F.binary_cross_entropy_with_logits(pl, torch.ones_like(pl)) + F.binary_cross_entropy_with_logits(nl, torch.zeros_like(nl))
        else:
            ref = F.softplus(nl - pl).mean()
        ref.backward()
        assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
        torch.testing.assert_close(gu, U.grad, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(ge, E.grad, rtol=1e-4, atol=1e-6)
        U.grad = E.grad = None
    z = np.load(os.path.join(G, "sasrec_bce.npz"))
    named, params = _sasrec_params(z, 2)
    seq = torch.from_numpy(z["in/seq"]).cuda()
    counts = torch.tensor([5, 7, 3])
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)[:-1]]).cuda()
    x = torch.stack([torch.randint(0, int(c), (16,), device="cuda", generator=g) for c in counts], 1)
    T = torch.randn(15, 8, device="cuda", generator=g).requires_grad_(True)
    TL = torch.randn(15, device="cuda", generator=g).requires_grad_(True)
    b = torch.zeros(1, device="cuda").requires_grad_(True)
    tests = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(R.sasrec_encoder.default, (named["Item.embeddings.weight"], named["Position.weight"], seq, params, 8.0, 0.0, 0), test_utils=tests)
    torch.library.opcheck(R.bce_pair.default, (U, E, pos, neg, valid, 0, 1), test_utils=tests)
    torch.library.opcheck(R.fm_bag.default, (T, TL, b, offsets, x), test_utils=tests)
