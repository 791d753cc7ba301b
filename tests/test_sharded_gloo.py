"""CPU, world_size 2, gloo: the row-sharded table's all-to-all routing with the oracle as local ops."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleLocalOps:
    """Test-only local kernels (CPU oracle) injected into ShardedTable so the exchange logic runs under gloo."""

    def gather(self, W, idx):
        from oracle import embedding
        return torch.from_numpy(embedding.gather_rows(W.numpy(), idx.numpy()))

    def scatter_add(self, g, idx, R):
        from oracle import ranking
        return torch.from_numpy(ranking.scatter_add_rows_c(g.numpy(), idx.numpy(), R))

    def score_topk(self, Q, E, seen_ptr, seen_idx, K):
        from oracle import ranking
        v, i = ranking.score_topk(Q.numpy(), E.numpy(), None if seen_ptr is None else seen_ptr.numpy(),
                                  None if seen_idx is None else seen_idx.numpy(), K)
        return torch.from_numpy(v), torch.from_numpy(i)

    def sparse_adam(self, g, idx, W, m, v, step, lr, b1, b2, eps, wd, padding_idx=-1):
        from oracle import adam
        keep = idx.numpy() != padding_idx
        adam.sparse_adam_rows(W.numpy(), m.numpy(), v.numpy(), idx.numpy()[keep], g.numpy()[keep], step, lr, b1, b2, eps, wd)   # in place (shared memory)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import ranking
        from recboard_amd.sharded import ShardedTable
        torch.manual_seed(0)                       # same "global" table on every rank
        R, D, K = 1001, 64, 20
        full = torch.randn(R, D)
        w = 1.0 / torch.arange(1, R + 1)
        # without dedup every looked-up position travels; with it (the default) every DISTINCT row does -- same rows, same gradients
        plain = ShardedTable(R, D, local_ops=OracleLocalOps(), dedup=False)
        plain.init_from_full(full)
        gp = torch.Generator().manual_seed(100 + rank)
        idx_p = torch.multinomial(w, 300, replacement=True, generator=gp).reshape(6, 50)
        rows_p, route_p = plain.lookup(idx_p)
        assert torch.equal(rows_p, full[idx_p]) and route_p.inv is None and route_p.n == 300
        grad_p = torch.randn(6, 50, D, generator=gp)
        shard_grad_plain = plain.backward(grad_p, route_p)
        tab = ShardedTable(R, D, local_ops=OracleLocalOps())
        tab.init_from_full(full)
        assert tab.local_rows == len(range(rank, R, world))
        g = torch.Generator().manual_seed(100 + rank)   # different batch on every rank
        idx = torch.multinomial(w, 300, replacement=True, generator=g).reshape(6, 50)   # Zipf: hot rows collide
        rows, route = tab.lookup(idx)
        assert torch.equal(rows, full[idx])                                               # == W[idx], bit exact
        assert route.n == len(torch.unique(idx)) < 300                                    # only distinct rows travelled
        # backward: every rank contributes gradient rows; owner's dense shard gradient == global scatter restricted
        grad = torch.randn(6, 50, D, generator=g)
        shard_grad = tab.backward(grad, route)
        all_idx = [torch.empty_like(idx) for _ in range(world)]
        all_grad = [torch.empty_like(grad) for _ in range(world)]
        dist.all_gather(all_idx, idx)
        dist.all_gather(all_grad, grad)
        ref = ranking.scatter_add_rows_c(torch.cat(all_grad).numpy(), torch.cat(all_idx).numpy(), R)
        np.testing.assert_allclose(shard_grad.numpy(), ref[rank::world], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(shard_grad_plain.numpy(), ref[rank::world], rtol=1e-5, atol=1e-5)
        # sparse optimizer step on the sharded table == sparse Adam on the unsharded table with everybody's gradient rows
        from oracle import adam as oadam
        Wref, mref, vref = full.numpy().copy(), np.zeros((R, D), np.float32), np.zeros((R, D), np.float32)
        oadam.sparse_adam_rows(Wref, mref, vref, torch.cat(all_idx).numpy().reshape(-1), torch.cat(all_grad).numpy().reshape(-1, D), 3, 1e-2, wd=1e-3)
        tab.backward_sparse_adam(grad, route, 3, 1e-2, weight_decay=1e-3)
        np.testing.assert_allclose(tab.weight.numpy(), Wref[rank::world], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(tab.m.numpy(), mref[rank::world], rtol=1e-5, atol=1e-6)
        tab.init_from_full(full)                   # (the scoring checks below use the original table)
        # sharded full-catalog top-K == unsharded oracle (bit-exact indices), with a seen mask
        Q = torch.randn(9, D, generator=g)
        seen = [np.unique(np.random.default_rng(rank * 10 + b).integers(0, R, 7)) for b in range(9)]
        sp = np.zeros(10, np.int64)
        sp[1:] = np.cumsum([len(x) for x in seen])
        si = np.concatenate(seen)
        v, i = tab.score_topk(Q, torch.from_numpy(sp), torch.from_numpy(si), K)
        rv, ri = ranking.score_topk(Q.numpy(), full.numpy(), sp, si, K)
        np.testing.assert_array_equal(i.numpy(), ri)
        np.testing.assert_array_equal(v.numpy(), rv)
        v, i = tab.score_topk(Q, None, None, K)
        rv, ri = ranking.score_topk(Q.numpy(), full.numpy(), None, None, K)
        np.testing.assert_array_equal(i.numpy(), ri)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def test_sharded_table_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"


def test_merge_topk_tie_rule():
    from recboard_amd.sharded import merge_topk
    v = torch.tensor([[3.0, 1.0, float("-inf"), 3.0, 2.0, 1.0]])
    i = torch.tensor([[7, 4, -1, 2, 9, 1]])
    mv, mi = merge_topk(v, i, 4)
    assert mi.tolist() == [[2, 7, 9, 1]] and mv.tolist() == [[3.0, 3.0, 2.0, 1.0]]


def test_counter_initialised_table_is_the_same_for_every_sharding():
    """The config-5 table is generated shard by shard from (seed, global row, column): any rank count sees the same values."""
    from recboard_amd.large import counter_normal_rows
    R, D, seed = 1003, 16, 5
    full = counter_normal_rows(torch.arange(R), D, seed, 0.02, "cpu")
    assert full.shape == (R, D) and torch.isfinite(full).all()
    assert abs(float(full.std()) - 0.02) < 1e-3 and abs(float(full.mean())) < 1e-3
    for G in (2, 3, 8):
        for r in range(G):
            rows = torch.arange(r, R, G)
            assert torch.equal(counter_normal_rows(rows, D, seed, 0.02, "cpu"), full[r::G])
    assert not torch.equal(counter_normal_rows(torch.arange(R), D, seed + 1, 0.02, "cpu"), full)
