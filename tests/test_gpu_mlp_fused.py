"""GPU: the fused launches of the DeepFM MLP step (DeepFM/main.py:119-124, 151-164, 201-215, 264-268) against float64 restatements of the
same expressions, and against the unfused entry points they replace.

  re_mlp_head_fwd / _bwd / _bwd_gated     the last Linear(., 1) + logit sum + BCELoss4Logits, and its backward
  re_gemm_f32_gated + re_bn_bwd_apply     the backward of dropout(relu(bn(.))) from the producing product's epilogue
  re_bn_relu_drop_fwd_pre                 the forward of the same block from the producing product's column partials
  re_grad_clip_coef + re_adam_step_scaled clip_grad_norm_(.., 10) + Adam
Tolerances: fp32 sums of 4 096 terms in a different order than torch's -> 2e-5 of the tensor's largest entry (written at each check)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from recboard_amd import ops as o
    return o


def close(a, b, tol=2e-5):
    a, b = a.double().cpu(), b.double().cpu()
    scale = float(b.abs().max()) or 1.0
    err = float((a - b).abs().max())
    assert err <= tol * scale, (err, scale)


def _block(ops, M, N, p, seed=3):
    g = torch.Generator().manual_seed(seed)
    z = (torch.randn(M, N, generator=g) * 1.5 + 0.3).cuda()
    gamma, beta = (torch.rand(N, generator=g) + 0.5).cuda(), (torch.randn(N, generator=g) * 0.1).cuda()
    rm, rv = torch.zeros(N).cuda(), torch.ones(N).cuda()
    a, stats = ops.bn_relu_drop_fwd(z, gamma, beta, rm, rv, True, p, seed=17, stream_id=101)
    return z, gamma, beta, a, stats


def _bn_bwd64(da, a, z, gamma, stats, p):
    M, N = z.shape
    ds = 1.0 / (1.0 - p) if p > 0 else 1.0
    da, a, z, gamma, stats = (t.double() for t in (da, a, z, gamma, stats))
    g = torch.where(a > 0, da * ds, torch.zeros_like(da))
    xhat = (z - stats[:N]) * stats[N:]
    dbeta, dgamma = g.sum(0), (g * xhat).sum(0)
    dz = stats[N:] * gamma * (g - dbeta / M - xhat * dgamma / M)
    return dz, dgamma, dbeta


@pytest.mark.parametrize("M,N,K,p", [(4096, 400, 400, 0.2), (4096, 400, 400, 0.0), (256, 112, 64, 0.5), (1024, 100, 400, 0.1)])
def test_gated_product_and_apply_match_the_three_launch_backward(ops, M, N, K, p):
    z, gamma, beta, a, stats = _block(ops, M, N, p)
    g0 = torch.Generator().manual_seed(5)
    dzn, W = (torch.randn(M, K, generator=g0) * 1e-3).cuda(), (torch.randn(K, N, generator=g0) * 0.05).cuda()
    da = (dzn.double() @ W.double())
    want_dz, want_dg, want_db = _bn_bwd64(da, a, z, gamma, stats, p)
    r = ops.gemm_gated(dzn, W, a, z, stats, 1.0 / (1.0 - p) if p > 0 else 1.0)
    assert r is not None
    g, part = r
    assert part.shape == (M // 64, 2, N)
    gref = torch.where(a > 0, da * (1.0 / (1.0 - p) if p > 0 else 1.0), torch.zeros_like(da))
    close(g, gref)
    # (dropped / inactive entries are exact zeros)
    assert float(g[a <= 0].abs().max() if bool((a <= 0).any()) else 0.0) == 0.0
    dgm, dbt = torch.empty(N).cuda(), torch.empty(N).cuda()
    dz = ops.bn_bwd_apply(g, z, gamma, stats, part, dgm, dbt)
    close(dgm, want_dg); close(dbt, want_db); close(dz, want_dz)
    # the path it replaces: product, then re_bn_relu_drop_bwd
    dz2, dg2, db2 = ops.bn_relu_drop_bwd(ops.gemm(dzn, W), a, z, gamma, stats, p)
    close(dz, dz2); close(dgm, dg2); close(dbt, db2)


def test_gated_product_refuses_what_the_wide_form_cannot_take(ops):
    z, gamma, beta, a, stats = _block(ops, 100, 48, 0.0)
    dzn, W = torch.randn(100, 64).cuda(), torch.randn(64, 48).cuda()
    assert ops.gemm_gated(dzn, W, a, z, stats, 1.0) is None        # M not a multiple of 64


@pytest.mark.parametrize("M,K,p", [(4096, 400, 0.2), (300, 16, 0.0), (64, 400, 0.5)])
def test_head_forward_backward_and_gated_backward(ops, M, K, p):
    z, gamma, beta, h, stats = _block(ops, M, K, p, seed=11)
    g0 = torch.Generator().manual_seed(2)
    w, b = (torch.randn(K, generator=g0) * 0.1).cuda(), torch.randn(1, generator=g0).cuda()
    fm_lr = torch.randn(M, generator=g0).cuda()
    y = (torch.rand(M, generator=g0) < 0.4).float().cuda()
    # forward + criterion
    logits = ops.mlp_head_fwd(h, w, b, fm_lr)
    want = h.double() @ w.double() + b.double() + fm_lr.double()
    close(logits, want, 1e-6)
    ds1, ds2 = torch.zeros(1).cuda(), torch.zeros(1).cuda()
    lg, loss, dl, dsum = ops.mlp_head_fwd(h, w, b, fm_lr, y, dsum=ds1, dsum2=ds2)
    assert torch.equal(lg, logits)
    want_loss = torch.nn.functional.binary_cross_entropy_with_logits(want, y.double())
    want_dl = (torch.sigmoid(want) - y.double()) / M
    assert abs(float(loss) - float(want_loss)) <= 2e-6 * max(1.0, float(want_loss))
    close(dl, want_dl, 2e-6)
    assert float(ds1) == float(ds2) == float(dsum)
    assert abs(float(ds1) - float(want_dl.sum())) <= 1e-6
    # backward, plain
    dW = torch.empty(K).cuda()
    da = ops.mlp_head_bwd(dl, h, w, dW)
    close(da, dl.double()[:, None] * w.double()[None, :], 1e-6)
    close(dW, dl.double() @ h.double())
    # backward, gated for the block underneath + the apply pass (the third column sum is dW)
    g, part = ops.mlp_head_bwd_gated(dl, h, w, z, stats, p)
    assert part.shape[1] == 3
    want_dz, want_dg, want_db = _bn_bwd64(dl.double()[:, None] * w.double()[None, :], h, z, gamma, stats, p)
    dgm, dbt, dW2 = torch.empty(K).cuda(), torch.empty(K).cuda(), torch.empty(K).cuda()
    dz = ops.bn_bwd_apply(g, z, gamma, stats, part, dgm, dbt, extra_out=dW2)
    close(dz, want_dz); close(dgm, want_dg); close(dbt, want_db); close(dW2, dl.double() @ h.double())
    # the criterion's sums deferred to the gated backward's launch (defer_final): the same bits as the two-launch forward, and the same g / part
    ds3, ds4 = torch.full((1,), 9.0).cuda(), torch.full((1,), 9.0).cuda()
    lg2, loss2, dl2, dsum2_, fin = ops.mlp_head_fwd(h, w, b, fm_lr, y, dsum=ds3, dsum2=ds4, defer_final=True)
    assert torch.equal(lg2, logits) and torch.equal(dl2, dl) and float(ds3) == 9.0          # (not written yet)
    g2, part2 = ops.mlp_head_bwd_gated(dl2, h, w, z, stats, p, final=fin)
    g_ref, part_ref = ops.mlp_head_bwd_gated(dl, h, w, z, stats, p)       # (bn_bwd_apply above turned the first g into dz in place)
    assert torch.equal(g2, g_ref) and torch.equal(part2, part_ref)
    assert torch.equal(loss2, loss) and float(ds3) == float(ds1) and float(ds4) == float(ds1)


@pytest.mark.parametrize("M,N,K", [(4096, 400, 100), (8192, 112, 64)])
def test_block_forward_from_the_products_partials_matches_the_two_pass_statistics(ops, M, N, K):
    g0 = torch.Generator().manual_seed(9)
    x, W, bias = torch.randn(M, K, generator=g0).cuda(), (torch.randn(N, K, generator=g0) * 0.2).cuda(), torch.randn(N, generator=g0).cuda()
    gamma, beta = (torch.rand(N, generator=g0) + 0.5).cuda(), (torch.randn(N, generator=g0) * 0.1).cuda()
    r = ops.gemm_colstats(x, W, True, bias)
    assert r is not None
    z, cs = r
    rm1, rv1, rm2, rv2 = torch.zeros(N).cuda(), torch.ones(N).cuda(), torch.zeros(N).cuda(), torch.ones(N).cuda()
    a1, s1 = ops.bn_relu_drop_fwd(z, gamma, beta, rm1, rv1, True, 0.3, seed=5, stream_id=100, colstats=cs)
    a2, s2 = ops.bn_relu_drop_fwd(z, gamma, beta, rm2, rv2, True, 0.3, seed=5, stream_id=100)
    zd = z.double()
    mean, var = zd.mean(0), zd.var(0, unbiased=False)
    close(s1[:N], mean, 1e-6); close(s1[N:], 1.0 / torch.sqrt(var + 1e-5), 1e-6)
    close(s1, s2, 1e-6); close(rm1, rm2, 1e-6); close(rv1, rv2, 1e-6)
    close(rv1, 0.9 + 0.1 * zd.var(0, unbiased=True), 1e-6)
    assert torch.equal(a1 == 0, a2 == 0) or float(((a1 == 0) != (a2 == 0)).float().mean()) < 1e-5   # (same dropout words; relu edges may flip)
    close(a1, a2, 1e-5)


def test_clip_coefficient_and_scaled_adam_match_torch(ops):
    g0 = torch.Generator().manual_seed(4)
    n = 100004          # (whole float4s: the arenas are padded to them)
    p0, gr = torch.randn(n, generator=g0), torch.randn(n, generator=g0) * 0.2
    for max_norm in (10.0, 1e9):
        pt = torch.nn.Parameter(p0.clone().double())
        pt.grad = gr.clone().double()
        opt = torch.optim.Adam([pt], lr=1e-2, weight_decay=1e-3)
        p, g, m, v = p0.clone().cuda(), gr.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
        for step in (1, 2, 3):
            pt.grad = (gr * step).clone().double()
            nrm = torch.nn.utils.clip_grad_norm_([pt], max_norm)
            opt.step()
            g.copy_(gr * step)
            coef = ops.grad_clip_coef(g, max_norm)
            assert abs(float(coef[1]) - float(nrm)) <= 1e-5 * float(nrm)
            assert abs(float(coef[0]) - min(1.0, max_norm / (float(nrm) + 1e-6))) <= 1e-6
            ops.adam_step_scaled(p, g, m, v, coef, step=step, lr=1e-2, weight_decay=1e-3)
            close(g, pt.grad, 1e-6)          # the gradient is left clipped, as p.grad is
            close(p, pt.data, 1e-6)


def test_bag_forward_hands_out_the_scatter_rows(ops):
    g0 = torch.Generator().manual_seed(1)
    counts = [7, 100, 3, 50]
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(counts)[:-1]]), dtype=torch.int64).cuda()
    R, D, B = sum(counts), 10, 333
    T, TL, bias = torch.randn(R, D, generator=g0).cuda(), torch.randn(R, generator=g0).cuda(), torch.zeros(1).cuda()
    x = torch.stack([torch.randint(0, c, (B,), generator=g0) for c in counts], 1).cuda()
    rows = torch.empty(B * len(counts), dtype=torch.int64).cuda()
    E, _ = ops.fm_bag_fwd(T, TL, bias, offsets, x, rows_out=rows)
    assert torch.equal(rows, (x + offsets[None, :]).reshape(-1))
    assert torch.equal(E, T[rows].reshape(B, -1))
