"""GPU: the row-sharded SASRec step on TWO ranks with the HIP kernels (re_route_bucket's real output, the bucket-order batch-local table,
the D = 128 compact-row step, the slot-map gather of the gradient rows, the row-sparse and dense Adam with device-side step scalars and the
overflow gate) against the unsharded CPU oracle.  Both ranks share this box's one GPU, so the collectives cannot be RCCL's (one rank per
device there): the process group is gloo and the two collectives the engine calls are staged through the host for this test -- every
kernel on the data path, and the exchange logic between two real ranks, is the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stage_collectives_through_the_host():
    a2a, ar = dist.all_to_all_single, dist.all_reduce

    def all_to_all_single(out, inp, out_splits=None, in_splits=None, group=None):
        if not out.is_cuda:
            return a2a(out, inp, out_splits, in_splits, group=group)
        o, i = torch.empty(out.shape, dtype=out.dtype), inp.detach().cpu().contiguous()
        a2a(o, i, out_splits, in_splits, group=group)
        out.copy_(o)

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None):
        if not t.is_cuda:
            return ar(t, op=op, group=group)
        c = t.detach().cpu()
        if op == dist.ReduceOp.AVG:                       # (gloo has no AVG)
            ar(c, op=dist.ReduceOp.SUM, group=group)
            c /= dist.get_world_size(group)
        else:
            ar(c, op=op, group=group)
        t.copy_(c)
    dist.all_to_all_single, dist.all_reduce = all_to_all_single, all_reduce


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        _stage_collectives_through_the_host()
        from oracle import adam as oadam, sasrec as osas
        from recboard_amd.large import SASRecShardedEngine, counter_normal_rows
        N, B, S, D, L, lr, wd = 997, 8, 50, 128, 2, 1e-2, 1e-4
        for factor, overflows in ((2.0, False), (0.05, True), ("default", None), (None, False)):
            eng = SASRecShardedEngine(N, S, D, L, dropout_rate=0.0, loss="BCE", lr=lr, weight_decay=wd, seed=3, device="cuda:0",
                                      capacity_factor=factor)
            rngs = [np.random.default_rng(50 + r) for r in range(world)]
            batches = []
            for r in range(world):
                seq = rngs[r].integers(1, N + 1, (B, S))
                for b in range(B):
                    seq[b, : rngs[r].integers(0, S - 1)] = 0
                pos = np.where(seq > 0, rngs[r].integers(0, N, (B, S)), 0)
                neg = np.where(seq > 0, rngs[r].integers(0, N, (B, S)), 0)
                batches.append(tuple(torch.from_numpy(a) for a in (seq, pos, neg)))
            # the unsharded reference on the host: the same counter-initialised table, loss = mean of the ranks' losses
            full = counter_normal_rows(torch.arange(N + 1), D, 3, 0.02, "cpu")
            full[0] = 0
            P = {k: p.detach().cpu().clone().requires_grad_(True) for k, p in eng.params.items()}
            Eref = full.clone().requires_grad_(True)
            P["Item.embeddings.weight"] = Eref
            losses = [osas.fit(P, *batches[r], "BCE", L) for r in range(world)]
            (sum(losses) / world).backward()
            mine = tuple(t.cuda() for t in batches[rank])
            loss = eng.train_step(*mine)
            n_over = eng.settle_overflow()
            if overflows is not None:
                assert (n_over == 1) == overflows, (factor, n_over)
            if n_over == 0:
                np.testing.assert_allclose(float(loss), float(losses[rank].detach()), rtol=2e-5)
            assert eng.arena.step == 1
            # Adam's first step moves an entry by lr * g / (|g| + eps): entries whose gradient is rounding noise may land anywhere in +- lr
            for k in eng.params:
                ref = P[k].detach().numpy().copy()
                g = P[k].grad.numpy() if P[k].grad is not None else np.zeros_like(ref)
                oadam.adam_step(ref, g, np.zeros_like(ref), np.zeros_like(ref), 1, lr, 0.9, 0.999, 1e-8, wd)
                got = eng.params[k].detach().cpu().numpy()
                solid = np.abs(g + wd * P[k].detach().numpy()) > 1e-5
                np.testing.assert_allclose(got[solid], ref[solid], rtol=1e-4, atol=2e-6, err_msg=k)
                assert np.abs(got - ref).max() <= 2.0 * lr + 1e-6, k
            rows = torch.nonzero(Eref.grad.abs().sum(1)).reshape(-1).numpy()
            rows = rows[rows != 0]
            Wref, m, v = full.numpy().copy(), np.zeros((N + 1, D), np.float32), np.zeros((N + 1, D), np.float32)
            oadam.sparse_adam_rows(Wref, m, v, rows, Eref.grad.numpy()[rows], 1, lr, wd=wd)
            got = eng.table.weight.cpu().numpy()
            want = Wref[rank::world]
            gfull = Eref.grad.numpy() + wd * full.numpy()
            solid = np.abs(gfull[rank::world]) > 1e-5
            np.testing.assert_allclose(got[solid], want[solid], rtol=1e-4, atol=2e-6)
            assert np.abs(got - want).max() <= 2.0 * lr + 1e-6
            untouched = np.ones(N + 1, bool)
            untouched[rows] = False
            np.testing.assert_array_equal(got[untouched[rank::world]], full.numpy()[rank::world][untouched[rank::world]])     # no other row moved
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def test_sharded_step_two_ranks_hip_kernels_match_unsharded_oracle():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"
