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
        ix = idx.numpy()
        ok = (ix >= 0) & (ix < W.shape[0])                       # the engine's rule: an out-of-range index reads a zero row
        out = embedding.gather_rows(W.numpy(), np.where(ok, ix, 0))
        out[~ok] = 0.0
        return torch.from_numpy(out)

    def scatter_add(self, g, idx, R):
        from oracle import ranking
        ix = idx.numpy().reshape(-1)
        ok = (ix >= 0) & (ix < R)                                # ... and an out-of-range destination is dropped
        return torch.from_numpy(ranking.scatter_add_rows_c(g.numpy().reshape(ix.size, -1)[ok], ix[ok], R))

    def route_bucket(self, idx, R, G, cap, skip_row=-1):
        """numpy restatement of re_route_bucket (csrc/route.hip): stable counting sort by owner, fixed capacity per peer."""
        ix = idx.numpy().reshape(-1)
        buckets = np.full((G, cap), -1, np.int64)
        slot = np.full(ix.size, -1, np.int64)
        counts = np.zeros(G + 1, np.int32)
        for j, r in enumerate(ix.tolist()):
            if r == skip_row:
                continue
            if r < 0 or r >= R:
                counts[G] += 1
                continue
            g = r % G
            k = counts[g]
            counts[g] += 1
            if k < cap:
                buckets[g, k] = r // G
                slot[j] = g * cap + k
            else:
                counts[G] += 1
        return torch.from_numpy(buckets), torch.from_numpy(slot), torch.from_numpy(counts)

    def score_topk(self, Q, E, seen_ptr, seen_idx, K):
        from oracle import ranking
        v, i = ranking.score_topk(Q.numpy(), E.numpy(), None if seen_ptr is None else seen_ptr.numpy(),
                                  None if seen_idx is None else seen_idx.numpy(), K)
        return torch.from_numpy(v), torch.from_numpy(i)

    def sparse_adam(self, g, idx, W, m, v, step, lr, b1, b2, eps, wd, padding_idx=-1):
        from oracle import adam
        keep = idx.numpy() != padding_idx
        adam.sparse_adam_rows(W.numpy(), m.numpy(), v.numpy(), idx.numpy()[keep], g.numpy()[keep], step, lr, b1, b2, eps, wd)   # in place (shared memory)

    def sparse_adam_dev(self, g, idx, W, m, v, hyper, b1, b2, eps, wd, padding_idx=-1):
        from oracle import adam
        keep = idx.numpy() != padding_idx
        adam.sparse_adam_rows(W.numpy(), m.numpy(), v.numpy(), idx.numpy()[keep], g.numpy()[keep], 0, 0.0, b1, b2, eps, wd, hyper=hyper.numpy())


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
        # ---- the sync-free fixed-capacity exchange (owner bucketing on the device, equal-split all-to-alls): same rows, same step
        fx = ShardedTable(R, D, local_ops=OracleLocalOps(), capacity_factor=2.0)
        fx.init_from_full(full)
        rows_f, route_f = fx.lookup(idx)
        assert torch.equal(rows_f, full[idx]) and route_f.cap == 300 and route_f.slot.numel() == 300
        fx.check_capacity()
        np.testing.assert_allclose(fx.backward(grad, route_f).numpy(), ref[rank::world], rtol=1e-5, atol=1e-5)
        fx.backward_sparse_adam(grad, route_f, 3, 1e-2, weight_decay=1e-3)
        np.testing.assert_allclose(fx.weight.numpy(), Wref[rank::world], rtol=1e-5, atol=1e-6)
        tight = ShardedTable(R, D, local_ops=OracleLocalOps(), capacity_factor=0.5)     # 75 slots per peer: the Zipf batch overflows
        tight.init_from_full(full)
        tight.lookup(idx)
        try:
            tight.check_capacity()
            raise AssertionError("the overflow went unnoticed")
        except RuntimeError as e:
            assert "capacity" in str(e)
        # ---- gather-on-save: the whole table on rank 0 from the shards (and back)
        got = fx.gather_full(0, chunk_rows=200)
        if rank == 0:
            np.testing.assert_allclose(got.numpy(), Wref, rtol=1e-5, atol=1e-6)
        else:
            assert got is None
        fx.load_full(full)
        assert torch.equal(fx.weight, full[rank::world])
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


# ---- the whole sharded SASRec training step (recboard_amd/large.py: SASRecShardedEngine.train_step) under gloo with two ranks: the
#      engine's orchestration (lookup -> batch-local table -> gradients -> contribution rows to their owners -> row-sparse Adam on
#      the shards, dense gradients averaged) with the ORACLE as compute, against the unsharded oracle step on the global batch.
def _engine_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import types
        from oracle import adam as oadam, sasrec as osas
        from recboard_amd.large import SASRecShardedEngine, counter_normal_rows

        class OracleShardedEngine(SASRecShardedEngine):
            """Same train_step; prepare_batch / _grads / _dense_adam (HIP kernels in the product) are the CPU oracle here."""

            def prepare_batch(self, seq, pos, neg):
                v = (seq != 0).reshape(-1)
                return types.SimpleNamespace(valid=v.to(torch.uint8), count=v.sum().to(torch.int32).reshape(1), plan=None,
                                             rows_all=torch.cat([seq.reshape(-1), torch.where(v, pos.reshape(-1) + 1, 0), torch.where(v, neg.reshape(-1) + 1, 0)]))

            def _grads(self, seq, pos, neg, aux, sd, seed_dev=None, table=None):
                P = {k: p.detach().clone().requires_grad_(True) for k, p in self.params.items()}
                T = table.clone().requires_grad_(True)
                P["Item.embeddings.weight"] = T
                loss = osas.fit(P, seq, pos, neg, self.loss_kind, self.L)
                loss.backward()
                for k, p in self.params.items():
                    self.arena.view(self.arena.grad, k).copy_(P[k].grad if P[k].grad is not None else torch.zeros_like(p))
                C = T.grad[1:].clone()                              # row 1 + j of the batch-local table = lookup j
                if not self.compact_form:
                    return loss.detach().reshape(1), C, None        # one contribution row per lookup, in lookup order
                # the product's compact-row form: only the rows that exist, in any order, tagged with their row of the batch-local
                # table (0 = belongs to nobody), padded with junk rows under key 0
                live = torch.nonzero(C.abs().sum(1)).reshape(-1)
                live = live[torch.randperm(live.numel(), generator=torch.Generator().manual_seed(4))]
                junk = torch.full((5, C.shape[1]), 7.0)
                return (loss.detach().reshape(1), torch.cat([C[live[:3]], junk, C[live[3:]]]),
                        torch.cat([live[:3] + 1, torch.zeros(5, dtype=torch.int64), live[3:] + 1]))

            def _dense_adam(self, hyper=None):
                A = self.arena
                oadam.adam_step(A.data.numpy(), A.grad.numpy(), A.m.numpy(), A.v.numpy(), A.step, self.lr, self.betas[0], self.betas[1], 1e-8, self.wd,
                                hyper=None if hyper is None else hyper.numpy())

        N, B, S, D, L, lr, wd = 97, 6, 50, 64, 2, 1e-2, 1e-4
        from tests.test_sharded_gloo import OracleLocalOps as Ops
        for factor, dedup, compact in ((None, False, False), (2.0, False, False), (None, False, True), (None, True, True), (2.0, False, True),
                                       (0.7, False, True), (0.05, True, True), ("default", True, True)):
            # (0.7: fits only because the padding row's lookups take no slot; 0.05: overflows -- the step is a no-op on every rank and is re-run
            #  on the exact-size path when its count is read; "default": the engine's default exchange)
            eng = OracleShardedEngine(N, S, D, L, dropout_rate=0.0, loss="BCE", lr=lr, weight_decay=wd, seed=3, device="cpu", dedup=dedup,
                                      capacity_factor=factor, local_ops=Ops())
            eng.compact_form = compact
            rngs = [np.random.default_rng(50 + r) for r in range(world)]
            batches = []
            for r in range(world):
                seq = rngs[r].integers(1, N + 1, (B, S))
                for b in range(B):
                    seq[b, : rngs[r].integers(0, S - 1)] = 0
                pos, neg = rngs[r].integers(0, N, (B, S)), rngs[r].integers(0, N, (B, S))
                batches.append(tuple(torch.from_numpy(a) for a in (seq, pos, neg)))
            # the unsharded reference: full table (same counter-based values), loss = mean of the ranks' losses, dense Adam on the encoder
            # parameters, row-sparse Adam on the rows the global batch touches
            full = counter_normal_rows(torch.arange(N + 1), D, 3, 0.02, "cpu")
            full[0] = 0
            P = {k: p.detach().clone().requires_grad_(True) for k, p in eng.params.items()}
            Eref = full.clone().requires_grad_(True)
            P["Item.embeddings.weight"] = Eref
            losses = [osas.fit(P, *batches[r], "BCE", L) for r in range(world)]
            (sum(losses) / world).backward()
            loss = eng.train_step(*batches[rank])
            if factor == 0.05:
                assert eng.overflow_steps == 0 and len(eng._pending) == 1          # (lag: not looked at yet; nothing has moved)
                np.testing.assert_array_equal(eng.table.weight.numpy(), full.numpy()[rank::world])
                assert eng.settle_overflow() == 1 and eng.arena.step == 1
            elif eng.settle_overflow() == 0:   # ("default" = 0.3 is sized for ~15 % real tokens; these batches are half real: either way the result is the same)
                np.testing.assert_allclose(float(loss), float(losses[rank].detach()), rtol=1e-6)
            else:
                assert factor == "default"
            if factor is not None:
                eng.table.check_capacity()
            A = eng.arena
            for k in eng.params:
                ref = P[k].detach().numpy().copy()
                g = P[k].grad.numpy() if P[k].grad is not None else np.zeros_like(ref)
                oadam.adam_step(ref, g, np.zeros_like(ref), np.zeros_like(ref), 1, lr, 0.9, 0.999, 1e-8, wd)
                # (Adam's first step is lr g / (|g| + eps): an entry whose gradient is rounding noise of the 4-rank all-reduce moves by a fraction of lr)
                np.testing.assert_allclose(eng.params[k].detach().numpy(), ref, rtol=1e-5, atol=1e-7 if world == 2 else 2e-4 * lr, err_msg=k)
            rows = torch.nonzero(Eref.grad.abs().sum(1)).reshape(-1).numpy()
            rows = rows[rows != 0]
            Wref, m, v = full.numpy().copy(), np.zeros((N + 1, D), np.float32), np.zeros((N + 1, D), np.float32)
            oadam.sparse_adam_rows(Wref, m, v, rows, Eref.grad.numpy()[rows], 1, lr, wd=wd)
            np.testing.assert_allclose(eng.table.weight.numpy(), Wref[rank::world], rtol=1e-5, atol=1e-7 if world == 2 else 2e-4 * lr)
            # gather-on-save through the engine: the reference's state-dict keys incl. the whole item table, on rank 0
            sd = eng.state_dict(0)
            if rank == 0:
                assert set(sd) == set(eng.params) | {"Item.embeddings.weight"}
                np.testing.assert_allclose(sd["Item.embeddings.weight"].numpy(), Wref, rtol=1e-5, atol=1e-7 if world == 2 else 2e-4 * lr)
            else:
                assert "Item.embeddings.weight" not in sd
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_engine_step_gloo_matches_unsharded_oracle(world):
    """world 2 and 4 (the ranks of `bench.py --gpus 4`'s config5_sharded leg, on a toy table): every exchange form, the overflowing one included
    -- there EVERY rank asserts that its step was a no-op until the count was read and was then re-run exactly once (settle_overflow() == 1)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_engine_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"
