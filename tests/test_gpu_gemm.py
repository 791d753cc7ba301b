"""GPU: general fp32 MFMA GEMM (all four transpose forms, ragged sizes, split-K, epilogues) vs float64 torch."""
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


@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (4096, 400, 100), (4096, 1, 400), (100, 400, 4096), (3000, 64, 12101),
                                    (12101, 64, 3000), (33, 17, 5), (257, 130, 67)])
@pytest.mark.parametrize("tA,tB", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_matches_float64(ops, M, N, K, tA, tB):
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N + 7 * K)
    A = torch.randn((K, M) if tA else (M, K), device="cuda", generator=g)
    B = torch.randn((N, K) if tB else (K, N), device="cuda", generator=g)
    out = ops.gemm(A, B, tA, tB)
    ref = ((A.T if tA else A).cpu().double() @ (B.T if tB else B).cpu().double())     # (on the host: no second GPU GEMM library in the comparison)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 2e-6 * K ** 0.5 * 10 + 1e-5, err
    assert torch.equal(out, ops.gemm(A, B, tA, tB))      # deterministic (split-K slabs are reduced in order)


def test_gemm_epilogues(ops):
    g = torch.Generator(device="cuda").manual_seed(0)
    x, W, b = torch.randn(300, 100, device="cuda", generator=g), torch.randn(40, 100, device="cuda", generator=g), torch.randn(40, device="cuda", generator=g)
    y = ops.gemm(x, W, transB=True, bias=b, relu=True)                       # relu(x W^T + b): an MLP layer
    torch.testing.assert_close(y, torch.relu(x @ W.T + b), rtol=1e-4, atol=1e-4)
    C0 = torch.randn(300, 40, device="cuda", generator=g)
    C = C0.clone()
    ops.gemm(x, W, transB=True, alpha=0.5, beta=2.0, out=C)
    torch.testing.assert_close(C, 0.5 * (x @ W.T) + 2.0 * C0, rtol=1e-4, atol=1e-4)
    # strided views (leading dimension > width)
    big = torch.randn(300, 256, device="cuda", generator=g)
    torch.testing.assert_close(ops.gemm(big[:, 16:116], W, transB=True), big[:, 16:116] @ W.T, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N,K,tA,tB", [(4096, 400, 100, False, True), (4096, 400, 400, False, True), (4096, 400, 400, False, False),
                                          (4096, 100, 400, False, False), (400, 400, 4096, True, False), (400, 100, 4096, True, False),
                                          (4096, 128, 64, False, True), (640, 48, 96, True, True), (4160, 404, 36, False, True)])
def test_wide_gemm_forms_of_the_deepfm_mlp(ops, M, N, K, tA, tB):
    """The wide form (csrc/gemm.hip: gemm_wide_k) at DeepFM's MLP shapes -- forward x W^T, input gradient dz W, weight gradient dz^T x (split K)
    -- and at ragged multiples of 4: against float64, deterministic, and with every epilogue."""
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N + 7 * K)
    A = torch.randn((K, M) if tA else (M, K), device="cuda", generator=g)
    B = torch.randn((N, K) if tB else (K, N), device="cuda", generator=g)
    bias = torch.randn(N, device="cuda", generator=g)
    ref = ((A.T if tA else A).cpu().double() @ (B.T if tB else B).cpu().double())
    out = ops.gemm(A, B, tA, tB)
    assert (out.cpu().double() - ref).abs().max().item() <= 2e-6 * K ** 0.5 * 10 + 1e-5
    assert torch.equal(out, ops.gemm(A, B, tA, tB))
    C0 = torch.randn(M, N, device="cuda", generator=g)
    C = C0.clone()
    ops.gemm(A, B, tA, tB, alpha=0.5, beta=2.0, out=C, bias=bias, relu=True)
    want = torch.relu(0.5 * ref + bias.cpu().double() + 2.0 * C0.cpu().double())
    assert (C.cpu().double() - want).abs().max().item() <= 2e-6 * K ** 0.5 * 10 + 1e-5


def test_gemm_colstats_are_the_batchnorm_statistics_of_the_output(ops):
    """re_gemm_f32_colstats: z = x W^T + b and, from the same launch, per-64-row (mean, M2) partials of z's columns that merge (Chan) into the
    batch mean / biased variance BatchNorm1d normalises with (DeepFM/main.py:119-124); re_bn_relu_drop_fwd_pre consumes them: the same
    activations and running statistics as the two-pass form."""
    g = torch.Generator(device="cuda").manual_seed(3)
    M, N, K = 4096, 400, 100
    x, W, b = torch.randn(M, K, device="cuda", generator=g) * 2 + 0.5, torch.randn(N, K, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
    z, cs = ops.gemm_colstats(x, W, True, b)
    z0 = ops.gemm(x, W, transB=True, bias=b)
    assert torch.equal(z, z0) and cs.shape == (M // 64, 2, N)
    zd = z.cpu().double().reshape(M // 64, 64, N)
    mean_b = zd.mean(1)
    torch.testing.assert_close(cs[:, 0].cpu().double(), mean_b, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(cs[:, 1].cpu().double(), ((zd - mean_b[:, None]) ** 2).sum(1), rtol=1e-4, atol=1e-3)
    gamma, beta = torch.rand(N, device="cuda", generator=g) + 0.5, torch.randn(N, device="cuda", generator=g)
    rm0, rv0 = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    rm1, rv1 = rm0.clone(), rv0.clone()
    a0, st0 = ops.bn_relu_drop_fwd(z, gamma, beta, rm0, rv0, True, 0.1, seed=7, stream_id=100)
    a1, st1 = ops.bn_relu_drop_fwd(z, gamma, beta, rm1, rv1, True, 0.1, seed=7, stream_id=100, colstats=cs)
    torch.testing.assert_close(st1, st0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a1, a0, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rm1, rm0, rtol=1e-5, atol=1e-6); torch.testing.assert_close(rv1, rv0, rtol=1e-5, atol=1e-6)
    full = z.cpu().double()
    torch.testing.assert_close(st1[:N].cpu().double(), full.mean(0), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(st1[N:].cpu().double(), 1.0 / torch.sqrt(full.var(0, unbiased=False) + 1e-5), rtol=1e-4, atol=1e-5)
    assert ops.gemm_colstats(x[:100], W, True, b) is None          # (M not a multiple of 64: the caller takes the two-pass form)


def test_deferred_split_k_reductions_give_the_same_bits(ops):
    """gemm(..., defer=pending) + gemm_reduce_many(pending) (re_gemm_f32_slabs + re_gemm_splitk_reduce_many: DeepFM's three weight-gradient
    products reduced by one launch) against the product that reduces itself: identical; a product that is not split is complete at once."""
    g = torch.Generator().manual_seed(4)
    pending, outs, refs = [], [], []
    for (M, N, K) in ((400, 400, 4096), (400, 100, 4096), (400, 400, 4096), (64, 64, 128)):
        A, B = torch.randn(K, M, generator=g).cuda(), torch.randn(K, N, generator=g).cuda()
        refs.append(ops.gemm(A, B, transA=True, alpha=0.5))
        out = torch.full((M, N), 7.0).cuda()
        ops.gemm(A, B, transA=True, alpha=0.5, out=out, defer=pending)
        outs.append(out)
    assert len(pending) == 3 and torch.equal(outs[3], refs[3])          # (the small one is not split: complete at once)
    ops.gemm_reduce_many(pending)
    assert pending == [] and all(torch.equal(o, r) for o, r in zip(outs, refs))
