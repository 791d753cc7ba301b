"""CPU, gloo, world 2 and 4: the data-parallel step with owner-computed Adam (recboard_amd/dp.py) against the UNSHARDED oracle --
dense Adam (oracle/adam.py, the reference's torch.optim.Adam) on the mean of every rank's gradient.  The per-rank compute is the numpy
restatement below (the HIP kernel re_adam_step_reduce is checked against the same restatement in tests/test_gpu_dp.py); what runs here is the
exchange: slices, all-to-all, all-gather, the padded form for an arena that does not split evenly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleReduceAdam:
    """Test-only local op: g = gscale * (((part 0 + part 1) + ...) in rank order), then the oracle's dense Adam -- in place on the views."""

    def reduce_adam(self, p, parts, m, v, step, lr, b1, b2, eps, wd, gscale, g_out=None, hyper=None):
        from oracle import adam
        g = parts[0].numpy().copy()
        for r in range(1, parts.shape[0]):
            g = g + parts[r].numpy()
        g = g * np.float32(gscale)
        if g_out is not None:
            g_out.numpy()[...] = g
        adam.adam_step(p.numpy(), g, m.numpy(), v.numpy(), step, lr, b1, b2, eps, wd)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, numel, q, preload=False):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import adam
        from recboard_amd.dp import OwnerAdam
        g0 = torch.Generator().manual_seed(7)                     # the same replica everywhere
        data = torch.randn(numel, generator=g0)
        grad, m, v = torch.zeros(numel), torch.zeros(numel), torch.zeros(numel)
        if preload:          # moments that exist before the hook does (a checkpoint loaded into the arena): every rank holds all of them
            m.copy_(torch.randn(numel, generator=g0) * 0.1)
            v.copy_(torch.rand(numel, generator=g0) * 0.01)
        ref_p, ref_m, ref_v = data.numpy().copy(), m.numpy().copy(), v.numpy().copy()
        dp = OwnerAdam(numel, local_ops=OracleReduceAdam(), device="cpu")
        assert dp.staged == (numel % (4 * world) != 0)
        assert dp.bytes_per_link_per_step == 2 * dp.chunk * 4
        lo, hi = dp.slice()
        lr, wd = 1e-2, 1e-3
        for step in ((6, 7, 8) if preload else (1, 2, 3)):
            gens = [torch.Generator().manual_seed(1000 * step + r) for r in range(world)]
            grads = [torch.randn(numel, generator=gg) * (1.0 + r) for r, gg in enumerate(gens)]      # every rank's own batch
            grad.copy_(grads[rank])
            dp.step(data, grad, m, v, step, lr, (0.9, 0.999), 1e-8, wd)
            # the unsharded step: mean gradient (summed in rank order), dense Adam
            gsum = grads[0].numpy().copy()
            for r in range(1, world):
                gsum = gsum + grads[r].numpy()
            gmean = gsum * np.float32(1.0 / world)
            adam.adam_step(ref_p, gmean, ref_m, ref_v, step, lr, 0.9, 0.999, 1e-8, wd)
            assert np.array_equal(data.numpy(), ref_p), f"rank {rank} step {step}: parameters differ from the unsharded oracle"
            assert np.array_equal(grad.numpy()[lo:hi], gmean[lo:hi])           # the owner's slice holds the averaged gradient
            assert np.array_equal(m.numpy()[lo:hi], ref_m[lo:hi]) and np.array_equal(v.numpy()[lo:hi], ref_v[lo:hi])
        # every replica holds the same bits
        all_p = [torch.empty_like(data) for _ in range(world)]
        dist.all_gather(all_p, data)
        assert all(torch.equal(all_p[0], t) for t in all_p)
        # a checkpoint's optimizer state: every owner's slice assembled on every rank
        fm, fv = dp.gather_moments(m, v)
        assert np.array_equal(fm.numpy(), ref_m) and np.array_equal(fv.numpy(), ref_v)
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world,numel,preload", [(2, 4096, False), (4, 4096, False), (2, 1003, False), (4, 8 * 999 + 4, False),
                                                 (2, 1003, True), (4, 4096, True), (4, 5, True)])     # (4, 5): ranks 2 and 3 own padding only
def test_owner_adam_step_matches_the_unsharded_oracle(world, numel, preload):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, numel, q, preload)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    bad = [r for r in res if r[1] != "ok"]
    assert not bad, bad
