"""N > 1 data-parallel replicas of one flat parameter arena: the optimizer step computed by the slice's OWNER  (SURVEY.md 8e).

The reference trains one replica (freerec/launcher.py's Coach; SASRec/main.py:264-275: loss.backward(); optimizer.step()); with N replicas of
the model, each on its own batch, the synchronous step is  g = mean over replicas of g_r,  then the reference's Adam on every replica.  Here:

    rank r owns the slice [r n / G, (r + 1) n / G) of the arena (parameters, gradient, Adam moments share one layout)
    1. all-to-all of the gradient arenas: rank r receives every rank's slice r                        (G - 1) n / G floats out, as many in
    2. ONE launch on the owner (re_adam_step_reduce): g = (1 / G) (((g_0 + g_1) + ...) in rank order), Adam on the slice -- a slice is
       reduced and updated exactly once, in a fixed order, so the replicas stay bit-identical (no drift between ranks, no dependence on a
       collective's reduction tree) and a rank touches 1 / G of the Adam traffic (22 MB -> 22 / G MB a step for the 3.3 MB SASRec arena)
    3. all-gather of the updated slices into every replica's parameter arena                            (G - 1) n / G floats out, as many in
    the Adam moments live on the owner only (1 / G of them per rank: `gather_moments` assembles them for a checkpoint).

Why slices and not "only the rows that changed": at the bench shape (B = 512 x S = 50, 12 101 items) a batch touches ~0.7 of the item
table's rows, and Adam with the reference's settings moves EVERY row every step (moment decay), so the replicas need every row back; a
row-sparse exchange pays off where rows touched << table rows -- that regime is the row-SHARDED table (recboard_amd/sharded.py, large.py:
indices out, rows back, gradient rows to the owner, row-sparse Adam there), not a replicated one.

Bytes: both collectives are one hop on a fully connected xGMI node (every pair of GPUs has its own link): per step a rank sends
2 (G - 1) n / G floats in total, (2 n / G) x 4 B over EACH of its G - 1 links (SASRec bench arena n = 0.83 M floats: G = 8 -> 0.83 MB per
link per step, 5.8 MB per rank), against the 2 (G - 1) sequential hops of a ring all-reduce of the same bytes.

`local_ops`: the per-rank compute (EngineLocalOps = the HIP kernel; tests inject a numpy restatement so the exchange runs under gloo).
"""
import torch
import torch.distributed as dist


class EngineLocalOps:
    def reduce_adam(self, p, parts, m, v, step, lr, b1, b2, eps, wd, gscale, g_out=None, hyper=None):
        from . import ops
        ops.adam_step_reduce(p, parts, m, v, step, lr, b1, b2, eps, wd, gscale=gscale, g_out=g_out, hyper=hyper)


class OwnerAdam:
    """owns_adam: an engine's `grad_hook` that also performs the optimizer step (SASRecEngine.train_step*, grad_hook=OwnerAdam(...))."""
    owns_adam = True

    def __init__(self, numel, group=None, local_ops=None, device="cuda"):
        self.group = group
        self.G = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.ops = local_ops if local_ops is not None else EngineLocalOps()
        self.numel = int(numel)
        q = 4 * self.G
        self.padded = (self.numel + q - 1) // q * q            # slices of whole float4s
        self.chunk = self.padded // self.G
        self.staged = self.padded != self.numel                # (an arena that does not split evenly goes through padded copies)
        self.recv = torch.empty(self.padded, dtype=torch.float32, device=device)
        if self.staged:
            self.gpad = torch.zeros(self.padded, dtype=torch.float32, device=device)
            self.ppad = torch.zeros(self.padded, dtype=torch.float32, device=device)
            self.mine = torch.zeros(3, self.chunk, dtype=torch.float32, device=device)     # the owner's (p, m, v) slice
        self.inplace_gather = dist.get_backend(group) == "nccl"
        self.bytes_out_per_step = 2 * (self.G - 1) * self.chunk * 4
        self.bytes_per_link_per_step = 2 * self.chunk * 4 if self.G > 1 else 0

    def slice(self):
        """The owned range [lo, hi) of the arena (empty -- lo == hi -- for a rank whose slice lies wholly in the padding)."""
        lo = min(self.rank * self.chunk, self.numel)
        return lo, max(lo, min(self.rank * self.chunk + self.chunk, self.numel))

    def step(self, data, grad, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, hyper=None):
        """One synchronous data-parallel optimizer step over the flat arenas (data, grad, m, v: [numel] each).  On return `data` holds the
        updated parameters of every slice; grad[own slice] the averaged gradient; m / v[own slice] the owner's moments."""
        G, c, lo = self.G, self.chunk, self.rank * self.chunk
        if G == 1 and not self.staged:
            # one replica: there is nobody to exchange with -- the owner's launch over the whole arena IS the step (no collective is issued
            # or recorded: a world-1 `--dp owner` step costs what the plain step costs; round 5 paid +18 % for two self-copies)
            self.ops.reduce_adam(data, grad.view(1, c), m, v, step, lr, betas[0], betas[1], eps, weight_decay, 1.0, g_out=None, hyper=hyper)
            return
        send = grad
        if self.staged:
            self.gpad[:self.numel].copy_(grad)
            send = self.gpad
        dist.all_to_all_single(self.recv, send, group=self.group)                  # equal splits: `chunk` floats per pair
        parts = self.recv.view(G, c)
        if not self.staged:
            p, mm, vv, gout, full = data[lo:lo + c], m[lo:lo + c], v[lo:lo + c], grad[lo:lo + c], data
        else:
            # the caller's arenas are the truth for the owner's slice, moments included (a loaded checkpoint, moments accumulated before this
            # hook existed): they go into the staging slice before the launch and come back behind it
            slo, hi = self.slice()
            p, mm, vv = self.mine[0], self.mine[1], self.mine[2]
            p[:hi - slo].copy_(data[slo:hi])
            mm[:hi - slo].copy_(m[slo:hi]); vv[:hi - slo].copy_(v[slo:hi])
            gout, full = self.gpad[lo:lo + c], self.ppad
        self.ops.reduce_adam(p, parts, mm, vv, step, lr, betas[0], betas[1], eps, weight_decay, 1.0 / G, g_out=gout, hyper=hyper)
        dist.all_gather_into_tensor(full, p if (self.inplace_gather or self.staged) else p.clone(), group=self.group)
        if self.staged:
            data.copy_(self.ppad[:self.numel])
            grad[slo:hi].copy_(self.gpad[slo:hi])
            m[slo:hi].copy_(mm[:hi - slo]); v[slo:hi].copy_(vv[:hi - slo])

    def step_arena(self, A, lr, betas, eps, weight_decay):
        """`A`: a ParamArena whose `step` the caller has already advanced (the engines' convention)."""
        self.step(A.data, A.grad, A.m, A.v, A.step, lr, betas, eps, weight_decay)

    def gather_moments(self, m, v):
        """-> (m, v) with every owner's slice in place on every rank (a checkpoint's optimizer state)."""
        out = []
        for t in (m, v):
            pad = torch.zeros(self.padded, dtype=torch.float32, device=t.device)
            lo, hi = self.slice()
            mine = torch.zeros(self.chunk, dtype=torch.float32, device=t.device)
            mine[:hi - lo].copy_(t[lo:hi])
            dist.all_gather_into_tensor(pad, mine, group=self.group)
            out.append(pad[:self.numel].clone())
        return out
