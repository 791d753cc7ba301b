"""Row-sharded embedding table for one node: row r lives on rank r % G as local row r // G.

`mod` placement spreads Zipf-hot rows over all GPUs.  Per step there is ONE exchange per direction:
    forward   indices out (all_to_all of int64 local row ids) -> owners gather -> rows back (all_to_all of fp32 rows)
    backward  gradient rows to their owners (all_to_all) -> local deterministic scatter-add into the shard's gradient
xGMI is point-to-point (7 links x ~153 GB/s per GPU): an all-to-all puts each pair's bytes on its own link, so the
exchange is bounded by the largest per-pair message, not by a ring.  The table's gradient is never all-reduced.

The local kernels are injected (`local_ops`): the product default is the HIP engine (recboard_amd.ops, GPU only, no
CPU fallback); the CPU `gloo` tests inject the oracle's gather / scatter-add to exercise the routing logic.
Full-catalog scoring shards the same way: queries are all-gathered (B*D*4 bytes -- tiny), every rank scores its shard
with the fused top-K kernel, the G partial lists are all-gathered and merged.
"""
import torch
import torch.distributed as dist


class EngineLocalOps:
    """HIP kernels (recboard_amd.ops).  Requires a GPU; raises otherwise."""

    def gather(self, W, idx):
        from . import ops
        return ops.gather_rows(W, idx)

    def scatter_add(self, g, idx, R):
        from . import ops
        return ops.scatter_add_rows(g, idx, R)

    def score_topk(self, Q, E, seen_ptr, seen_idx, K, prep=None):
        from . import ops
        return ops.score_topk(Q, E, seen_ptr, seen_idx, K, prep=prep)

    def score_prepare(self, E):
        """The shard's bf16 planes for the split scoring path (None where there is no split form): built once per
        ShardedTable.score_topk call -- the shard is scored against every rank's queries."""
        from . import ops
        return ops.score_prepare(E)

    def sparse_adam(self, g, idx, W, m, v, step, lr, b1, b2, eps, wd, padding_idx=-1):
        from . import ops
        if ops.sparse_adam_small_ok(idx, W):       # a step's worth of rows: one launch, no sort (re_sparse_adam_rows_small)
            ops.sparse_adam_rows_small(g, idx.reshape(-1), W, m, v, step=step, lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, padding_idx=padding_idx)
        else:
            ops.sparse_adam_rows(g, idx, W, m, v, step, lr, b1, b2, eps, wd, padding_idx=padding_idx)

    def sparse_adam_dev(self, g, idx, W, m, v, hyper, b1, b2, eps, wd, padding_idx=-1):
        """sparse_adam with the step-dependent scalars in device memory (captured steps)."""
        from . import ops
        if ops.sparse_adam_small_ok(idx, W):
            ops.sparse_adam_rows_small(g, idx.reshape(-1), W, m, v, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, padding_idx=padding_idx, hyper=hyper)
        else:
            ops.sparse_adam_rows_dev(g, idx, W, m, v, hyper, b1, b2, eps, wd, padding_idx=padding_idx)

    def route_bucket(self, idx, R, G, cap, skip_row=-1):
        """Owner bucketing with a fixed capacity per peer (re_route_bucket): no host sync."""
        from . import ops
        return ops.route_bucket(idx, R, G, cap, skip_row=skip_row)


class Route:
    """Permutation + split sizes of one lookup, kept for the backward exchange.  dedup: `inv` maps every looked-up position to
    its distinct row among the `n` rows that travel.  Fixed-capacity form: `slot` (position of every lookup in the [G, cap] buckets),
    `cap`, and `recv_local` = the [G * cap] local row ids the peers want from this rank (-1 = unused slot)."""
    __slots__ = ("order", "send_counts", "recv_counts", "recv_local", "n", "inv", "slot", "cap", "dropped")


class ShardedTable:
    def __init__(self, num_rows, dim, local_ops=None, group=None, device=None, dtype=torch.float32, dedup=True, capacity_factor=None,
                 skip_row=None):
        # dedup: every distinct row of a batch crosses the fabric once per direction -- the indices out, the
        # rows back, and in the backward ONE pre-summed gradient row per distinct index (Zipf-distributed lookups repeat their
        # hot rows many times: config 2's 76 800 lookups per step hit ~12 000 distinct rows)
        self.dedup = dedup
        # capacity_factor c: the exchange takes the FIXED-CAPACITY form -- every peer pair moves ceil(c * n / G) slots per
        # direction (equal-split all-to-alls, sizes known without looking at the data: no host sync, capturable), the lookups are
        # bucketed by owner on the device (re_route_bucket), unused slots travel as -1 / zero rows.  With mod placement a factor of 2
        # holds unless one rank owns > 2/G of a batch's lookups; c = G can never overflow.  Lookups that do not fit are counted in
        # `self.dropped` (device int32, summed over calls): the caller checks it at its next sync point.  None: exact split sizes
        # through the host (two syncs per lookup), optionally with dedup.
        self.capacity_factor = capacity_factor
        # skip_row (fixed-capacity form): lookups of this global row -- the padding row, ~85 % of a left-padded SASRec batch -- are not
        # exchanged at all (slot -1: a zero row comes back, no gradient row leaves), so the capacity is sized for the real tokens
        self.skip_row = -1 if skip_row is None else int(skip_row)
        self.dropped = None
        self.group = group
        self.G = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.R, self.D = num_rows, dim
        self.local_rows = (num_rows - self.rank + self.G - 1) // self.G
        self.ops = local_ops if local_ops is not None else EngineLocalOps()
        self.device = device
        self.weight = torch.zeros((self.local_rows, dim), dtype=dtype, device=device)

    # ---- placement
    def owner(self, idx):
        return idx % self.G

    def local_index(self, idx):
        return idx // self.G

    def global_index(self, local, rank=None):
        return local * self.G + (self.rank if rank is None else rank)

    def init_from_full(self, full):
        """Test helper: take this rank's rows of a replicated table."""
        self.weight.copy_(full[self.rank::self.G])

    # ---- exchange
    def _route(self, idx):
        flat = idx.reshape(-1)
        inv = None
        if self.dedup:
            flat, inv = torch.unique(flat, return_inverse=True)
        own = self.owner(flat)
        order = torch.argsort(own, stable=True)
        send_counts = torch.bincount(own, minlength=self.G)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()       # split sizes are host-side by API
        send_local = self.local_index(flat[order]).contiguous()
        recv_local = torch.empty(sum(rc), dtype=flat.dtype, device=flat.device)
        dist.all_to_all_single(recv_local, send_local, rc, sc, group=self.group)
        r = Route()
        r.order, r.send_counts, r.recv_counts, r.recv_local, r.n, r.inv, r.slot, r.cap = order, sc, rc, recv_local, flat.numel(), inv, None, 0
        r.dropped = None
        return r

    def _route_fixed(self, idx):
        flat = idx.reshape(-1).contiguous()
        n, G = flat.numel(), self.G
        cap = max(1, min(n, -(-int(self.capacity_factor * n) // G)))
        if self.skip_row >= 0:
            buckets, slot, counts = self.ops.route_bucket(flat, self.R, G, cap, self.skip_row)
        else:
            buckets, slot, counts = self.ops.route_bucket(flat, self.R, G, cap)
        # every bucket carries one more word: this rank's count of lookups that found no slot -- after the exchange every rank holds
        # all G counts, so "did ANY rank overflow in this lookup" is known everywhere without a collective of its own (route.dropped)
        send = torch.cat([buckets, counts[G:G + 1].to(buckets.dtype).expand(G).unsqueeze(1)], 1).contiguous()
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv.view(-1), send.view(-1), group=self.group)               # equal splits: cap ids + 1 per pair
        if self.dropped is None:
            self.dropped = torch.zeros(1, dtype=counts.dtype, device=counts.device)
        self.dropped.add_(counts[G:G + 1])               # (in place: a captured step keeps counting across replays)
        r = Route()
        r.slot, r.cap, r.recv_local, r.n, r.inv, r.order, r.send_counts, r.recv_counts = slot, cap, recv[:, :cap].reshape(-1), n, None, None, None, None
        r.dropped = recv[:, cap].sum().reshape(1)        # the same number on every rank
        return r

    def check_capacity(self):
        """Host-side check (one sync) that no lookup since the last check was dropped by the fixed-capacity exchange.  (A caller that
        gates and re-runs overflowing steps itself -- SASRecShardedEngine -- sets `raise_on_overflow = False`.)"""
        if self.dropped is not None and getattr(self, "raise_on_overflow", True):
            d = int(self.dropped)
            self.dropped.zero_()
            if d:
                raise RuntimeError(f"ShardedTable: {d} lookups exceeded the exchange capacity (capacity_factor={self.capacity_factor}); "
                                   f"use capacity_factor={self.G} (never overflows) or None (exact sizes through the host)")

    def lookup(self, idx, expand=True, exact=False):
        """-> (rows [*idx.shape, D], route).  Global `W[idx]` on a table no rank holds entirely.
        exact=True: the exact-size exchange (split sizes through the host) even when the table has a capacity factor.
        expand=False (fixed-capacity form only): -> (table [1 + G * cap, D], route): a zero row followed by the received rows in BUCKET
        order -- lookup j's row is row `route.slot[j] + 1` (slot -1: the zero row); the caller indexes it instead of asking for a row
        per lookup."""
        if self.capacity_factor is not None and not exact:
            r = self._route_fixed(idx)
            rows_for_peers = self.ops.gather(self.weight, r.recv_local).reshape(-1, self.D)       # (-1 -> a zero row)
            if not expand:       # (received straight into rows 1.. of the caller's batch-local table; row 0 = the zero / padding row)
                table = torch.empty((1 + rows_for_peers.shape[0], self.D), dtype=rows_for_peers.dtype, device=rows_for_peers.device)
                table[0].zero_()
                dist.all_to_all_single(table[1:], rows_for_peers.contiguous(), group=self.group)
                return table, r
            rows_recv = torch.empty_like(rows_for_peers)
            dist.all_to_all_single(rows_recv, rows_for_peers.contiguous(), group=self.group)       # equal splits: cap rows per pair
            out = self.ops.gather(rows_recv, r.slot)
            return out.reshape(tuple(idx.shape) + (self.D,)), r
        r = self._route(idx)
        rows_for_peers = self.ops.gather(self.weight, r.recv_local).reshape(-1, self.D)
        rows_sorted = torch.empty((r.n, self.D), dtype=self.weight.dtype, device=self.weight.device)
        dist.all_to_all_single(rows_sorted, rows_for_peers.contiguous(), r.send_counts, r.recv_counts, group=self.group)
        out = torch.empty_like(rows_sorted)
        out[r.order] = rows_sorted
        if r.inv is not None:
            out = self.ops.gather(out, r.inv)          # distinct rows -> every looked-up position
        return out.reshape(tuple(idx.shape) + (self.D,)), r

    def backward(self, grad_rows, route, positions=None):
        """Send every gradient row to the owner of its table row; -> dense gradient of THIS rank's shard."""
        g = self._grad_rows_to_send(grad_rows, route, positions)
        recv = self._exchange_grad_rows(g, route)
        return self.ops.scatter_add(recv, route.recv_local, self.local_rows)

    def _exchange_grad_rows(self, g, route):
        if route.slot is not None:
            recv = torch.empty_like(g)
            dist.all_to_all_single(recv, g, group=self.group)                                       # equal splits
            return recv
        recv = torch.empty((sum(route.recv_counts), self.D), dtype=g.dtype, device=g.device)
        dist.all_to_all_single(recv, g, route.recv_counts, route.send_counts, group=self.group)
        return recv

    def _grad_rows_to_send(self, grad_rows, route, positions=None, slots=None):
        """positions (optional, int64 [m]): grad_rows[i] belongs to lookup positions[i] of the routed index list (-1: to nobody);
        slots (optional, fixed-capacity form): grad_rows[i] belongs to bucket slot slots[i] (rows of `lookup(expand=False)`; -1: nobody;
        every slot at most once);
        default: one gradient row per lookup, in lookup order."""
        g = grad_rows.reshape(-1, self.D)
        if slots is not None:
            # every bucket slot belongs to ONE lookup, so at most one row goes into it: no sums to form -- invert the map (integers only)
            # and gather the rows into bucket order (-1, a slot nobody feeds: a zero row)
            slots = slots.reshape(-1)
            src = torch.full((self.G * route.cap + 1,), -1, dtype=torch.int64, device=g.device)
            src[slots + 1] = torch.arange(slots.numel(), device=g.device)          # (entry 0 collects the slot -1 rows: dropped)
            return self.ops.gather(g.contiguous(), src[1:]).reshape(-1, self.D)
        if positions is not None:
            positions = positions.reshape(-1)
            live = positions >= 0
            at = positions.clamp_min(0)
        if route.slot is not None:                     # fixed capacity: into bucket order (slots are distinct; -1 = dropped)
            slot = route.slot if positions is None else torch.where(live, route.slot[at], torch.full_like(at, -1))
            return self.ops.scatter_add(g.contiguous(), slot, self.G * route.cap)
        if route.inv is not None:                      # one pre-summed row per distinct index (deterministic segmented sum)
            inv = route.inv if positions is None else torch.where(live, route.inv[at], torch.full_like(at, -1))
            g = self.ops.scatter_add(g.contiguous(), inv, route.n)
        elif positions is not None:                    # (no dedup: one row per lookup; lookups nobody contributes to send a zero row)
            g = self.ops.scatter_add(g.contiguous(), torch.where(live, at, torch.full_like(at, -1)), route.n)
        return g[route.order].contiguous()

    def backward_sparse_adam(self, grad_rows, route, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, padding_global_row=None,
                             positions=None, hyper=None, slots=None, gate=None):
        """Training form for tables whose dense gradient does not fit: gradient rows go to their owners (the same single
        all-to-all as `backward`) and the owner applies ONE row-sparse Adam update per distinct row of its shard (summed
        duplicates, SparseAdam rule; moments `m`, `v` live next to the shard).  No table-sized gradient ever exists.
        gate (device bool [1], optional): True turns the update into a no-op (every received row id becomes -1: "nobody's")."""
        if not hasattr(self, "m"):
            self.m = torch.zeros_like(self.weight)
            self.v = torch.zeros_like(self.weight)
        g = self._grad_rows_to_send(grad_rows, route, positions, slots)
        recv = self._exchange_grad_rows(g, route)
        pad = -1
        if padding_global_row is not None and self.owner(padding_global_row) == self.rank:   # the padding row is never updated
            pad = self.local_index(padding_global_row)
        if gate is not None:
            route.recv_local = torch.where(gate, torch.full_like(route.recv_local, -1), route.recv_local)
        if hyper is not None:     # (captured step: step size and bias correction come from device memory, `step` / `lr` are ignored)
            self.ops.sparse_adam_dev(recv, route.recv_local, self.weight, self.m, self.v, hyper, betas[0], betas[1], eps, weight_decay, padding_idx=pad)
        else:
            self.ops.sparse_adam(recv, route.recv_local, self.weight, self.m, self.v, step, lr, betas[0], betas[1], eps, weight_decay, padding_idx=pad)

    # ---- checkpoints: the full table exists on no rank; gather-on-save / scatter-on-load go through rank `dst` in row chunks
    def gather_full(self, dst=0, tensor=None, chunk_rows=1 << 20):
        """-> the whole [R, D] table (or `tensor`, e.g. an Adam moment, laid out like the shard) on rank `dst`, None elsewhere.
        Every rank must call it.  The reference saves `model.state_dict()` with the whole nn.Embedding (ETEGRec/train_etegrec.py:549-574)."""
        W = self.weight if tensor is None else tensor
        full = torch.empty((self.R, self.D), dtype=W.dtype, device=W.device) if self.rank == dst else None
        maxl = (self.R + self.G - 1) // self.G
        for l0 in range(0, maxl, chunk_rows):
            n = min(chunk_rows, maxl - l0)
            mine = torch.zeros((n, self.D), dtype=W.dtype, device=W.device)
            k = max(0, min(n, self.local_rows - l0))
            mine[:k] = W[l0:l0 + k]
            parts = [torch.empty_like(mine) for _ in range(self.G)] if self.rank == dst else None
            dist.gather(mine, parts, dst=dist.get_global_rank(self.group, dst) if self.group is not None else dst, group=self.group)
            if self.rank == dst:
                for g in range(self.G):
                    rows = torch.arange(l0, l0 + n, device=W.device) * self.G + g
                    ok = rows < self.R
                    full[rows[ok]] = parts[g][: int(ok.sum())]
        return full

    def load_full(self, full, tensor=None):
        """Inverse of gather_full for a table every rank can read (e.g. a checkpoint loaded on the host): take this rank's rows."""
        (self.weight if tensor is None else tensor).copy_(torch.as_tensor(full)[self.rank::self.G].to(self.weight.device))

    # ---- full-catalog scoring over the sharded catalog
    def score_topk(self, Q_local, seen_ptr, seen_idx, K):
        """Q_local [b, D] on every rank; seen CSR in GLOBAL item ids for the local queries.
        -> (vals [b, K], idx [b, K] global ids), identical to scoring against the unsharded table."""
        G, dev = self.G, Q_local.device
        b = Q_local.shape[0]
        nb = torch.tensor([b], dtype=torch.int64, device=dev)
        nbs = [torch.empty_like(nb) for _ in range(G)]
        dist.all_gather(nbs, nb, group=self.group)
        if any(int(x) != b for x in nbs):
            raise ValueError(f"ShardedTable.score_topk: every rank must pass the same number of queries (got {[int(x) for x in nbs]}); "
                             "pad the last evaluation batch")
        Qs = [torch.empty_like(Q_local) for _ in range(G)]
        dist.all_gather(Qs, Q_local.contiguous(), group=self.group)
        ptrs, idxs = self._gather_seen(seen_ptr, seen_idx)
        outs_v, outs_i = [], []
        prep = self.ops.score_prepare(self.weight) if hasattr(self.ops, "score_prepare") else None
        kw = {} if prep is None else {"prep": prep}
        for src in range(G):                 # score every rank's queries against MY shard
            sp, si = self._seen_for_shard(ptrs[src], idxs[src])
            v, i = self.ops.score_topk(Qs[src], self.weight, sp, si, min(K, self.local_rows), **kw)
            gi = torch.where(i >= 0, self.global_index(i), i)
            if v.shape[1] < K:               # shard smaller than K: pad
                pad = K - v.shape[1]
                v = torch.cat([v, torch.full((b, pad), float("-inf"), device=dev)], 1)
                gi = torch.cat([gi, torch.full((b, pad), -1, dtype=gi.dtype, device=dev)], 1)
            outs_v.append(v)
            outs_i.append(gi)
        # every rank ends up with the G partial lists of ITS queries
        sv, si_ = torch.stack(outs_v).contiguous(), torch.stack(outs_i).contiguous()      # [G, b, K], equal splits
        pv, pi = torch.empty_like(sv), torch.empty_like(si_)
        dist.all_to_all_single(pv, sv, group=self.group)
        dist.all_to_all_single(pi, si_, group=self.group)
        return merge_topk(pv.permute(1, 0, 2).reshape(b, G * K), pi.permute(1, 0, 2).reshape(b, G * K), K)

    def _gather_seen(self, seen_ptr, seen_idx):
        G = self.G
        if seen_ptr is None:
            return [None] * G, [None] * G
        n = torch.tensor([seen_idx.numel()], dtype=torch.int64, device=seen_idx.device)
        ns = [torch.empty_like(n) for _ in range(G)]
        dist.all_gather(ns, n, group=self.group)
        ptrs = [torch.empty_like(seen_ptr) for _ in range(G)]
        dist.all_gather(ptrs, seen_ptr.contiguous(), group=self.group)
        mx = int(max(int(x) for x in ns))
        padded = torch.full((mx,), -1, dtype=seen_idx.dtype, device=seen_idx.device)
        padded[: seen_idx.numel()] = seen_idx
        idxs = [torch.empty_like(padded) for _ in range(G)]
        dist.all_gather(idxs, padded, group=self.group)
        return ptrs, [x[: int(k)] for x, k in zip(idxs, ns)]

    def _seen_for_shard(self, ptr, idx):
        """Keep the seen ids owned by this rank, as LOCAL ids (ascending order is preserved)."""
        if ptr is None:
            return None, None
        mine = self.owner(idx) == self.rank
        counts = torch.zeros(ptr.numel() - 1, dtype=torch.int64, device=idx.device)
        rows = torch.repeat_interleave(torch.arange(ptr.numel() - 1, device=idx.device), ptr[1:] - ptr[:-1])
        counts.index_add_(0, rows[mine], torch.ones(int(mine.sum()), dtype=torch.int64, device=idx.device))
        sp = torch.zeros_like(ptr)
        sp[1:] = torch.cumsum(counts, 0)
        return sp, self.local_index(idx[mine]).contiguous()


def merge_topk(vals, idx, K):
    """K best of concatenated partial lists [b, G*K]; order = value descending, ties -> lowest index; (-inf, -1) last."""
    key_idx = torch.where(idx < 0, torch.full_like(idx, torch.iinfo(idx.dtype).max), idx)
    o1 = torch.argsort(key_idx, dim=1, stable=True)
    v1, i1 = torch.gather(vals, 1, o1), torch.gather(idx, 1, o1)
    o2 = torch.argsort(v1, dim=1, descending=True, stable=True)
    return torch.gather(v1, 1, o2)[:, :K].contiguous(), torch.gather(i1, 1, o2)[:, :K].contiguous()
