"""GPU: DeepFM bag / FM / LR / BCE kernels and the DeepFM engine vs the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _engine(z):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from recboard_amd.deepfm import DeepFMEngine
    counts = z["cfg/counts"].tolist()
    m = DeepFMEngine(counts, 10, (32, 24, 16), batch_norm=True, hidden_dropout_rate=0.0, lr=1e-3)
    for f, (t, tl) in enumerate(zip(m.tables(), m.tables_lr())):
        t.copy_(dev(z[f"table/{f}"]))
        tl.copy_(dev(z[f"table_lr/{f}"]))
    m.bias.copy_(dev(z["param/fm.lr_layer.bias"]))
    m.load_dnn_state_dict({k[len("param/dnn."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/dnn.")})
    return m


def test_fm_bag_and_bce_kernels_vs_oracle():
    from oracle import criterions
    z = np.load(os.path.join(G, "deepfm.npz"))
    m = _engine(z)
    from recboard_amd import ops
    x = dev(z["in/x"])
    E, fm_lr = ops.fm_bag_fwd(m.T, m.TL.reshape(-1), m.bias, m.offsets, x)
    tabs = [torch.from_numpy(z[f"table/{f}"]) for f in range(10)]
    tl = [torch.from_numpy(z[f"table_lr/{f}"]) for f in range(10)]
    xc = torch.from_numpy(z["in/x"])
    Er = torch.stack([tabs[f][xc[:, f]] for f in range(10)], 1)
    np.testing.assert_array_equal(E.cpu().numpy(), Er.flatten(1).numpy())
    fm = 0.5 * (Er.sum(1) ** 2 - (Er ** 2).sum(1)).sum(-1)
    lr = torch.stack([tl[f][xc[:, f]] for f in range(10)], 1).sum(1).squeeze(-1) + torch.from_numpy(z["param/fm.lr_layer.bias"])
    np.testing.assert_allclose(fm_lr.cpu().numpy(), (fm + lr).numpy(), rtol=1e-5, atol=1e-6)
    logits = torch.randn(32)
    labels = torch.from_numpy(z["in/labels"]).float().reshape(-1)
    lg = logits.clone().requires_grad_(True)
    ref = criterions.bce_with_logits(lg, labels)
    ref.backward()
    loss, dl, ds = ops.bce_logits(logits.cuda(), labels.cuda())
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-6)
    np.testing.assert_allclose(dl.cpu().numpy(), lg.grad.numpy(), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(ds.item(), lg.grad.sum().item(), rtol=1e-4, atol=1e-7)


def test_deepfm_engine_matches_reference_golden():
    z = np.load(os.path.join(G, "deepfm.npz"))
    m = _engine(z).train()
    x, y = dev(z["in/x"]), dev(z["in/labels"])
    logits, _ = m.encode(x)
    np.testing.assert_allclose(logits.cpu().numpy(), z["out/train_logits"].reshape(-1), rtol=1e-4, atol=1e-5)
    m2 = _engine(z).train()
    loss = m2.forward_backward(x, y)
    np.testing.assert_allclose(loss.item(), float(z["out/rec_loss"]), rtol=1e-5)
    for f, (o, c) in enumerate(zip(m2.offsets.tolist(), m2.counts)):
        np.testing.assert_allclose(m2.gT[o:o + c].cpu().numpy(), z[f"gtable/{f}"], rtol=1e-3, atol=2e-6)
        np.testing.assert_allclose(m2.gTL[o:o + c].cpu().numpy(), z[f"gtable_lr/{f}"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(m2.gbias.cpu().numpy(), z["grad/fm.lr_layer.bias"], rtol=1e-4, atol=1e-7)
    for k in z.files:          # every MLP gradient: Linear weights/biases, BatchNorm affine
        if k.startswith("grad/dnn."):
            ref = z[k]
            got = m2.G[k[5:]].cpu().numpy().reshape(ref.shape)
            assert np.abs(got - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-3) + 1e-7, k
    for i in range(3):         # running statistics updated like nn.BatchNorm1d (momentum 0.1, unbiased variance)
        np.testing.assert_allclose(m2.running[i][0].cpu().numpy(), z[f"post/dnn.{i}.bn.running_mean"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(m2.running[i][1].cpu().numpy(), z[f"post/dnn.{i}.bn.running_var"], rtol=1e-4, atol=1e-6)
    # eval pass: running stats as updated by the train-mode forward
    m2.eval()
    np.testing.assert_allclose(m2.recommend_from_pool(x).cpu().numpy(), z["out/eval_scores"], rtol=1e-4, atol=1e-6)


def test_deepfm_train_step_runs_and_learns():
    z = np.load(os.path.join(G, "deepfm.npz"))
    m = _engine(z).train()
    x, y = dev(z["in/x"]), dev(z["in/labels"])
    losses = [m.train_step(x, y).item() for _ in range(40)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0] - 0.05


def test_captured_step_replays_the_eager_step():
    """DeepFMEngine.train_step_graph: the same seeds (device words), the same launches -- identical parameters, moments and running statistics."""
    from recboard_amd.deepfm import DeepFMEngine
    counts = [40, 30, 7, 3]
    rng = np.random.default_rng(6)
    B = 96
    batches = [(torch.from_numpy(np.stack([rng.integers(0, c, B) for c in counts], 1)).cuda(), torch.from_numpy((rng.random((B, 1)) < 0.4).astype(np.int64)).cuda())
               for _ in range(4)]
    a = DeepFMEngine(counts, 8, (32, 16), batch_norm=True, hidden_dropout_rate=0.2, lr=1e-2, seed=5)
    b = DeepFMEngine(counts, 8, (32, 16), batch_norm=True, hidden_dropout_rate=0.2, lr=1e-2, seed=5)
    for x, y in batches:
        la = a.train_step(x, y).clone()
        lb = b.train_step_graph(x, y).clone()
        assert torch.equal(la.reshape(-1), lb.reshape(-1)), (la, lb)
    assert a.step == b.step == 4
    assert torch.equal(a.data, b.data) and torch.equal(a.m, b.m) and torch.equal(a.v, b.v)
    for i in a.running:
        assert torch.equal(a.running[i][0], b.running[i][0]) and torch.equal(a.running[i][1], b.running[i][1])


@pytest.mark.parametrize("B,D,counts", [(4096, 10, [94762, 25612, 7, 24, 12, 5, 50, 500, 5000, 50000]), (1000, 10, [3, 1, 40000]),
                                        (8192, 15, [2, 300000]), (33, 4, [5, 6, 7, 8, 9]), (1, 10, [10, 10])])
def test_fm_table_grad_matches_index_add(B, D, counts):
    """re_fm_table_grad (csrc/fmbag.hip: a workgroup per field sorts the field's keys in LDS and sums the runs) against index_add in fp64:
    heavy rows (a field of 2 - 7 values takes B / 2 ... B / 7 contributions a row), rows nobody refers to stay as the caller left them (zero),
    B that is no power of two, one key; and the same bits on a second call."""
    from recboard_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    F, R = len(counts), sum(counts)
    off = torch.tensor([sum(counts[:i]) for i in range(F)], dtype=torch.int64, device="cuda")
    x = torch.stack([torch.randint(0, c, (B,), device="cuda", generator=g) for c in counts], 1)
    rows = (x + off).contiguous()
    kt = x.t().contiguous().to(torch.int32)
    gE = torch.randn(B * F, D, device="cuda", generator=g)
    gL = torch.randn(B * F, 1, device="cuda", generator=g)
    assert ops.fm_table_grad_ok(B, F, D, max(counts))
    gT, gTL = torch.zeros(R, D, device="cuda"), torch.zeros(R, device="cuda")
    sl = ops.fm_table_slices(counts, B, "cuda")
    ops.fm_table_grad(kt, off, sl, gE, gL, gT, gTL)
    refT = torch.zeros(R, D, device="cuda", dtype=torch.float64).index_add_(0, rows.reshape(-1), gE.double())
    refL = torch.zeros(R, device="cuda", dtype=torch.float64).index_add_(0, rows.reshape(-1), gL.reshape(-1).double())
    tol = 1e-6 * max(8.0, (B / min(counts)) ** 0.5 * 4)          # fp32 sums of up to B terms
    assert float((gT.double() - refT).abs().max()) <= tol * float(refT.abs().max() + 1)
    assert float((gTL.double() - refL).abs().max()) <= tol * float(refL.abs().max() + 1)
    untouched = torch.ones(R, dtype=torch.bool, device="cuda")
    untouched[rows.reshape(-1)] = False
    assert float(gT[untouched].abs().max() if untouched.any() else 0.0) == 0.0
    gT2, gTL2 = torch.zeros_like(gT), torch.zeros_like(gTL)
    ops.fm_table_grad(kt, off, sl, gE, gL, gT2, gTL2)
    assert torch.equal(gT, gT2) and torch.equal(gTL, gTL2)
    # any slicing is a correct one: one slice per field (every list long: the LDS sort and the stretch walk), and 3-row slices
    for other in (torch.tensor([(f, 0, c, 0) for f, c in enumerate(counts)], dtype=torch.int32, device="cuda"),
                  torch.tensor([(f, lo, min(lo + 3, c), 0) for f, c in enumerate(counts) for lo in range(0, c, 3)][:200000], dtype=torch.int32, device="cuda")):
        if other.shape[0] == 200000:
            continue
        gT3, gTL3 = torch.zeros_like(gT), torch.zeros_like(gTL)
        ops.fm_table_grad(kt, off, other, gE, gL, gT3, gTL3)
        assert float((gT3.double() - refT).abs().max()) <= tol * float(refT.abs().max() + 1)
        assert float((gTL3.double() - refL).abs().max()) <= tol * float(refL.abs().max() + 1)


def test_fm_table_grad_drops_keys_outside_the_tables():
    from recboard_amd import ops
    counts = [4, 6]
    off = torch.tensor([0, 4], dtype=torch.int64, device="cuda")
    rows = torch.tensor([[0, 4], [3, 9], [1, 10], [-1, 5]], dtype=torch.int64, device="cuda")      # 10: past the end; -1: in front of field 0
    kt = (rows - off).t().contiguous().to(torch.int32)
    gE, gL = torch.ones(8, 3, device="cuda"), torch.ones(8, 1, device="cuda")
    gT, gTL = torch.zeros(10, 3, device="cuda"), torch.zeros(10, device="cuda")
    ops.fm_table_grad(kt, off, ops.fm_table_slices(counts, 4, "cuda"), gE, gL, gT, gTL)
    assert gTL.tolist() == [1, 1, 0, 1, 1, 1, 0, 0, 0, 1]
    assert not ops.fm_table_grad_ok(8193, 2, 10, 6) and not ops.fm_table_grad_ok(64, 2, 16, 6) and not ops.fm_table_grad_ok(64, 2, 10, 1 << 19)
