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
