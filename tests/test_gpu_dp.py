"""GPU: the owner's launch of the data-parallel step (re_adam_step_reduce) against the numpy restatement the gloo tests use
(tests/test_dp_gloo.py), and recboard_amd.dp.OwnerAdam over RCCL with ONE rank inside the SASRec engine's captured step: it must take
exactly the steps of the plain engine (the exchange with 2 and 4 ranks is covered under gloo)."""
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_reduce_adam_kernel_matches_the_restatement():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import adam
    from recboard_amd import ops
    g0 = torch.Generator().manual_seed(3)
    n, G = 40004, 4
    p0 = torch.randn(n, generator=g0)
    parts = torch.randn(G, n + 8, generator=g0)[:, :n]              # (a row stride larger than n)
    for nparts in (1, 2, 4):
        p, m, v, gout = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda(), torch.empty(n).cuda()
        rp, rm, rv = p0.numpy().copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
        pd = parts.cuda()[:nparts]
        for step in (1, 2, 3):
            ops.adam_step_reduce(p, pd, m, v, step, 1e-2, weight_decay=1e-3, gscale=1.0 / nparts, g_out=gout)
            g = parts[0].numpy().copy()
            for r in range(1, nparts):
                g = g + parts[r].numpy()
            g = g * np.float32(1.0 / nparts)
            assert np.array_equal(gout.cpu().numpy(), g)            # the reduction: the same order, bit for bit
            adam.adam_step(rp, g, rm, rv, step, 1e-2, 0.9, 0.999, 1e-8, 1e-3)
            np.testing.assert_allclose(p.cpu().numpy(), rp, rtol=2e-6, atol=1e-7)     # (fp32 Adam: the kernel's fma contraction vs numpy)
            np.testing.assert_allclose(m.cpu().numpy(), rm, rtol=2e-6, atol=1e-9)
    # one part, scale 1: re_adam_step's arithmetic, bit for bit
    p, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    p2, m2, v2 = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    g = parts[0].contiguous().cuda()
    for step in (1, 2):
        ops.adam_step_reduce(p, g.view(1, n), m, v, step, 1e-2, weight_decay=1e-3)
        ops.adam_step(p2, g, m2, v2, step, 1e-2, weight_decay=1e-3)
        assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2)


def test_owner_adam_on_one_rank_takes_the_plain_engines_steps():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.distributed as dist
    from recboard_amd.dp import OwnerAdam
    from recboard_amd.sasrec import SASRecEngine
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    plain = dp_eng = hook = None
    try:
        N, B, S = 500, 16, 50
        rng = np.random.default_rng(5)
        kw = dict(dropout_rate=0.2, loss="BCE", lr=1e-2, weight_decay=1e-4, seed=4)
        plain, dp_eng = SASRecEngine(N, S, 64, 2, **kw), SASRecEngine(N, S, 64, 2, **kw)
        hook = OwnerAdam(dp_eng.arena.numel)
        assert hook.owns_adam and not hook.staged and hook.bytes_per_link_per_step == 0
        for step in range(4):
            seq = rng.integers(1, N + 1, (B, S))
            for b in range(B):
                seq[b, : rng.integers(0, S - 1)] = 0
            batch = tuple(torch.from_numpy(a).cuda() for a in (seq, rng.integers(0, N, (B, S)), rng.integers(0, N, (B, S))))
            if step < 2:
                l0, l1 = plain.train_step_graph(*batch).clone(), dp_eng.train_step_graph(*batch, grad_hook=hook).clone()
            else:
                l0, l1 = plain.train_step_fused(*batch).clone(), dp_eng.train_step_fused(*batch, grad_hook=hook).clone()
            assert torch.equal(l0, l1), step
            # (the plain engine's fused step runs Adam inside its tail launch: the same update on the same gradient)
            torch.testing.assert_close(dp_eng.arena.data, plain.arena.data, rtol=1e-6, atol=1e-8)
            torch.testing.assert_close(dp_eng.arena.m, plain.arena.m, rtol=1e-6, atol=1e-10)
    finally:
        # (the captured steps hold the communicator's launches: drop them before the group goes)
        del plain, dp_eng, hook
        import gc
        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()
