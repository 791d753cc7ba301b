"""CPU: the engine's operators are registered PyTorch custom ops (torch.ops.recengine.*) with fake (meta)
implementations, and have NO CPU kernel: a CPU tensor fails loudly instead of falling back."""
import pytest
import torch

from recboard_amd import torch_ops  # noqa: F401

R = torch.ops.recengine


def test_ops_are_registered_with_schemas():
    for name in ("gather_rows", "scatter_add_rows", "bpr_triplet", "score_dense", "score_topk", "spmm_csr"):
        op = getattr(R, name)
        assert "recengine::" + name in str(op.default._schema)


def test_fake_implementations_propagate_shapes():
    m = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device="meta")  # noqa: E731
    W, idx = m(100, 64), m(7, 50, dt=torch.int64)
    assert R.gather_rows(W, idx).shape == (7, 50, 64)
    assert R.scatter_add_rows(m(350, 64), m(350, dt=torch.int64), 100, 0).shape == (100, 64)
    loss, logits = R.bpr_triplet(W, W, m(32, dt=torch.int64), m(32, dt=torch.int64), m(32, dt=torch.int64))
    assert loss.shape == () and logits.shape == (32, 2)
    assert R.score_dense(m(8, 64), W).shape == (8, 100)
    v, i = R.score_topk(m(8, 64), W, m(9, dt=torch.int64), m(0, dt=torch.int64), 50)
    assert v.shape == (8, 50) and i.shape == (8, 50) and i.dtype == torch.int64
    assert R.spmm_csr(m(101, dt=torch.int64), m(400, dt=torch.int64), m(400), m(100, 64)).shape == (100, 64)


def test_no_cpu_kernel():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        R.gather_rows(torch.zeros(4, 8), torch.zeros(3, dtype=torch.long))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        R.score_topk(torch.zeros(4, 8), torch.zeros(5, 8), torch.zeros(5, dtype=torch.long), torch.zeros(0, dtype=torch.long), 3)
