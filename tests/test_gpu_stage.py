"""The launches around a captured step that round 6 fused: re_step_stage_inputs (batch copies + label cast + step scalars + zero fills in one
launch: csrc/adam.hip) and re_adam_step_clip2 (clip_grad_norm_ + Adam over two weight-decay groups behind the norm partials)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stage_inputs_copies_casts_zeroes_and_writes_the_scalars():
    from recboard_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randint(-5, 1 << 40, (4096, 10), device="cuda", generator=g)            # int64 batch
    y = torch.randint(0, 2, (4097,), device="cuda", generator=g)                       # int64 labels -> fp32 (odd length)
    f = torch.randn(1001, device="cuda", generator=g)                                  # fp32, length not a multiple of 4
    big = torch.randn(3_000_003, device="cuda", generator=g)
    odd = torch.randn(77, device="cuda", generator=g)[1:]                              # a 4-byte aligned, not 16-byte aligned view
    sx, sy, sf = torch.empty_like(x), torch.full((4097,), 9.0, device="cuda"), torch.empty_like(f)
    so = torch.full((76,), 7.0, device="cuda")
    z1, z2 = torch.full((1 << 20,), 3.0, device="cuda"), torch.full((13,), 3.0, device="cuda")[1:]
    state = torch.zeros(4, dtype=torch.int32, device="cuda")
    ops.stage_inputs(state, 0xDEADBEEF, 7, 1e-3, 0.9, 0.999, pairs=[(sx, x), (sy, y), (sf, f), (so, odd)], zeros=[z1, z2])
    assert torch.equal(sx, x) and torch.equal(sy, y.float()) and torch.equal(sf, f) and torch.equal(so, odd)
    assert float(z1.abs().max()) == 0.0 and float(z2.abs().max()) == 0.0
    st = state.cpu()
    assert st[0].item() & 0xFFFFFFFF == 0xDEADBEEF and st[1].item() == 0
    hyper = state.view(torch.float32)[2:4].cpu()
    assert math.isclose(hyper[0].item(), 1e-3 / (1 - 0.9 ** 7), rel_tol=1e-6) and math.isclose(hyper[1].item(), 1 / math.sqrt(1 - 0.999 ** 7), rel_tol=1e-6)
    # the same words as re_step_state leaves
    ref = torch.zeros(4, dtype=torch.int32, device="cuda")
    ops.step_state(ref, 0xDEADBEEF, 7, 1e-3, 0.9, 0.999)
    assert torch.equal(ref, state)
    # more than eight segments, a dtype pair the launch does not take, a non-contiguous input: the ordinary way, same result
    srcs = [torch.randn(5 + i, device="cuda") for i in range(10)]
    dsts = [torch.empty_like(s) for s in srcs]
    h = torch.randn(8, 6, device="cuda")
    dh, di = torch.empty(6, 8, device="cuda"), torch.empty(9, dtype=torch.float64, device="cuda")
    ops.stage_inputs(None, 0, 1, 1e-3, 0.9, 0.999, pairs=list(zip(dsts, srcs)) + [(dh, h.t()), (di, torch.arange(9, device="cuda"))], zeros=[big])
    assert all(torch.equal(a, b) for a, b in zip(dsts, srcs)) and torch.equal(dh, h.t()) and torch.equal(di, torch.arange(9, device="cuda").double())
    assert float(big.abs().max()) == 0.0


@pytest.mark.parametrize("scale", [0.01, 50.0])
def test_adam_step_clip2_is_clip_then_two_adam_groups(scale):
    """Against re_grad_clip_coef + two re_adam_step_scaled (the four launches it replaces): the same coefficient to fp32 rounding (the partials
    are added in another fixed order), hence parameters to ~1e-6; scale 0.01: the norm is under max_norm (coefficient exactly 1), 50: clipped."""
    from recboard_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    n, ne = 2_000_000, 1_200_000
    p0 = torch.randn(n, device="cuda", generator=g)
    g0 = torch.randn(n, device="cuda", generator=g) * scale / math.sqrt(n)
    m0, v0 = torch.randn(n, device="cuda", generator=g) * 1e-3, torch.rand(n, device="cuda", generator=g) * 1e-4
    pa, ga, ma, va = (t.clone() for t in (p0, g0, m0, v0))
    coef = ops.grad_clip_coef(ga, 10.0)
    ops.adam_step_scaled(pa[:ne], ga[:ne], ma[:ne], va[:ne], coef, step=5, lr=1e-3, weight_decay=0.05)
    ops.adam_step_scaled(pa[ne:], ga[ne:], ma[ne:], va[ne:], coef, step=5, lr=1e-3, weight_decay=0.0)
    pb, gb, mb, vb = (t.clone() for t in (p0, g0, m0, v0))
    out = ops.adam_step_clip2(pb, gb, mb, vb, ne, 10.0, step=5, lr=1e-3, wd_first=0.05, wd_rest=0.0)
    torch.testing.assert_close(out, coef, rtol=1e-6, atol=0)
    if scale < 1:
        assert float(out[0]) == 1.0 and torch.equal(gb, g0)
    else:
        assert float(out[0]) < 1.0
    for a, b in ((pa, pb), (ga, gb), (ma, mb), (va, vb)):
        torch.testing.assert_close(b, a, rtol=2e-6, atol=1e-9)
    # the device-word form (a captured step's) gives the same bits as the host-scalar form
    state = torch.zeros(4, dtype=torch.int32, device="cuda")
    ops.step_state(state, 1, 5, 1e-3, 0.9, 0.999)
    pc, gc, mc, vc = (t.clone() for t in (p0, g0, m0, v0))
    ops.adam_step_clip2(pc, gc, mc, vc, ne, 10.0, hyper=state.view(torch.float32)[2:4], wd_first=0.05, wd_rest=0.0)
    assert torch.equal(pc, pb) and torch.equal(mc, mb) and torch.equal(vc, vb) and torch.equal(gc, gb)


def test_captured_deepfm_and_mf_steps_are_reproducible_run_to_run():
    """Two engines each, the same seeds and batches, a dozen captured steps: parameters identical bit for bit -- through re_fm_table_grad
    (slices with one row and many keys, slices with many rows and few), the staging launch, the fused clip + Adam, and the owner launch's
    two-set form (a table beyond 256 x 96 rows)."""
    import numpy as np
    from recboard_amd.deepfm import DeepFMEngine
    from recboard_amd.gen import MFEngine
    rng = np.random.default_rng(2)
    counts, B = [30000, 7, 3, 900, 50], 1024
    bs = [(torch.from_numpy(np.stack([rng.integers(0, c, B) for c in counts], 1)).cuda(), torch.from_numpy((rng.random((B, 1)) < 0.3).astype(np.int64)).cuda())
          for _ in range(3)]
    outs = []
    for _ in range(2):
        m = DeepFMEngine(counts, 10, (64, 64), batch_norm=True, hidden_dropout_rate=0.1, lr=1e-3, embedding_decay=0.05, seed=1)
        for i in range(12):
            m.train_step_graph(*bs[i % 3])
        outs.append(m.data.clone())
    assert torch.equal(*outs)
    U, N, Bm = 20000, 9000, 512                       # 29 000 rows x 2 halves > 256 x 96: the wide owner form
    tb = [tuple(torch.from_numpy(rng.integers(0, n, Bm)).cuda() for n in (U, N, N)) for _ in range(3)]
    outs = []
    for _ in range(2):
        m = MFEngine(U, N, 64, lr=1e-3, weight_decay=1e-8, seed=1)
        for i in range(12):
            m.train_step_graph(*tb[i % 3])
        outs.append(m.arena.data.clone())
    assert torch.equal(*outs)
