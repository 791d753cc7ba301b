"""GPU: SASRec on the engine vs the reference's golden vectors (tests/golden/sasrec_*.npz) and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _engine(z, loss, **kw):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from recboard_amd.sasrec import SASRecEngine
    m = SASRecEngine(int(z["cfg/N"]), 50, int(z["cfg/D"]), int(z["cfg/num_blocks"]), dropout_rate=0.0, loss=loss, **kw)
    m.load_state_dict({k[6:]: z[k] for k in z.files if k.startswith("param/") and z[k].dtype == np.float32})
    return m


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("loss", ["BCE", "BPR"])
def test_fit_loss_and_grads_match_reference_golden(loss):
    z = np.load(os.path.join(G, f"sasrec_{loss.lower()}.npz"))
    m = _engine(z, loss)
    seq, pos, neg = dev(z["in/seq"]), dev(z["in/pos"]), dev(z["in/neg"])
    A = m.arena
    for k, p in m.params.items():
        p.grad = A.view(A.grad, k)
    L = m.fit(seq, pos, neg)["rec_loss"]
    np.testing.assert_allclose(L.item(), float(z["out/rec_loss"]), rtol=1e-5)
    L.backward()
    for k, p in m.params.items():
        ref = z["grad/" + k]
        scale = max(np.abs(ref).max(), 1e-6)
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err <= 1e-4 * scale + 1e-7, (k, err, scale)   # north_star tolerance: 1e-4 relative
    assert (m.params["Item.embeddings.weight"].grad[0] == 0).all()   # padding row gets no gradient


def test_encode_scores_topk_match_reference_golden():
    z = np.load(os.path.join(G, "sasrec_bce.npz"))
    m = _engine(z, "BCE").eval()
    seq = dev(z["in/seq"])
    with torch.no_grad():
        u, _ = m.encode(seq)
    np.testing.assert_allclose(u.cpu().numpy(), z["out/userEmbds"], rtol=1e-4, atol=2e-5)
    sc = m.recommend_from_full(seq)
    np.testing.assert_allclose(sc.cpu().numpy(), z["out/scores"], rtol=1e-4, atol=2e-5)
    vals, idx = m.recommend_topk(seq, dev(z["in/seen_ptr"]), dev(z["in/seen_idx"]), 50)
    np.testing.assert_array_equal(idx.cpu().numpy(), z["out/topk_idx"])     # bit-exact top-K indices
    np.testing.assert_allclose(vals.cpu().numpy(), z["out/topk_vals"], rtol=1e-4, atol=2e-5)


def test_train_step_matches_oracle_adam_trajectory():
    """3 full steps (zero_grad, backward, dense Adam with coupled L2) vs oracle fit + torch.optim.Adam on CPU."""
    from oracle import sasrec as osas
    z = np.load(os.path.join(G, "sasrec_bce.npz"))
    m = _engine(z, "BCE", lr=5e-4, weight_decay=1e-6)
    P = osas.params_from_npz(z, requires_grad=True)
    opt = torch.optim.Adam(list(P.values()), lr=5e-4, betas=(0.9, 0.999), weight_decay=1e-6)
    seq, pos, neg = (torch.from_numpy(z[k]) for k in ("in/seq", "in/pos", "in/neg"))
    for step in range(3):
        l_gpu = m.train_step(seq.cuda(), pos.cuda(), neg.cuda())
        opt.zero_grad()
        l_cpu = osas.fit(P, seq, pos, neg, "BCE", 2)
        l_cpu.backward()
        for p in P.values():       # params absent from the graph still take the dense L2 + moment update
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        opt.step()
        np.testing.assert_allclose(l_gpu.item(), l_cpu.item(), rtol=2e-5)
    for k, p in m.params.items():
        np.testing.assert_allclose(p.detach().cpu().numpy(), P[k].detach().numpy(), rtol=1e-3, atol=2e-5, err_msg=k)
