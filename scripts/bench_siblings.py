#!/usr/bin/env python3
"""Sibling models (recboard_amd.siblings) on the engine's torch custom ops vs the same module code on aten's operators
(index / einsum / F.linear / torch.sparse.mm -- what the reference's model files execute on a GPU), one training step each:
zero_grad, fit, backward, torch.optim.Adam.step.  The aten variants exist only here (the product has no second backend): the
module-level functions of recboard_amd.nn are swapped for the duration of the measurement.

    python scripts/bench_siblings.py [--steps 30] > gpurun_out/siblings.json
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recboard_amd import nn as rnn  # noqa: E402
from recboard_amd import siblings as sib  # noqa: E402

ENGINE = {k: getattr(rnn, k) for k in ("gather_rows", "linear", "spmm_sym", "spmm", "bpr_triplet", "score_full")}


def _aten_gather(W, idx, padding_idx=-1):
    return F.embedding(idx, W, padding_idx=(padding_idx if padding_idx >= 0 else None))


def _aten_bpr(Ut, It, users, pos, neg):
    u = Ut[users]
    return F.softplus((u * It[neg]).sum(-1) - (u * It[pos]).sum(-1)).mean()


_csr_cache = {}


def _aten_csr(crow, col, val, n):
    key = (crow.data_ptr(), col.data_ptr(), val.data_ptr())
    if key not in _csr_cache:
        _csr_cache[key] = torch.sparse_csr_tensor(crow, col, val, size=(n, n))
    return _csr_cache[key]


ATEN = {
    "gather_rows": _aten_gather,
    "linear": lambda x, w, b=None: F.linear(x, w, b),
    "spmm_sym": lambda crow, col, val, X, plan=None: torch.sparse.mm(_aten_csr(crow, col, val, X.shape[0]), X),
    "spmm": lambda A, At, X: torch.sparse.mm(_aten_csr(*A, X.shape[0]), X),
    "bpr_triplet": _aten_bpr,
    "score_full": lambda Q, E: Q @ E.t(),
}


def use(table):
    for k, f in table.items():
        setattr(rnn, k, f)


def bipartite(U, N, E, gen):
    u = torch.randint(0, U, (E,), generator=gen)
    i = (torch.rand(E, generator=gen) ** 2 * N).long().clamp_(max=N - 1)       # popularity-skewed items
    key = torch.unique(u * N + i)
    u, i = key // N, key % N
    return u, i


def sym_adj(U, N, u, i, self_loops=False, left=False):
    n = U + N
    r = torch.cat((u, i + U)); c = torch.cat((i + U, u))
    if self_loops:
        ar = torch.arange(n)
        r, c = torch.cat((r, ar)), torch.cat((c, ar))
    deg = torch.bincount(r, minlength=n).float().clamp_min(1)
    v = (1.0 / deg[r]) if left else (deg[r].rsqrt() * deg[c].rsqrt())
    order = torch.argsort(r * n + c)
    crow = torch.zeros(n + 1, dtype=torch.int64)
    crow[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
    return crow.cuda(), c[order].contiguous().cuda(), v[order].contiguous().cuda()


def timed(model, step, steps):
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)

    def one():
        opt.zero_grad(set_to_none=True)
        losses = step()
        sum(losses.values()).backward()
        opt.step()

    for _ in range(5):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def timed_graph(model, lossf, inputs, steps):
    """The same step as ONE hipGraph replay (nn.GraphedStep)."""
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3, capturable=True)
    step = rnn.GraphedStep(model, lossf, opt, inputs)
    for _ in range(5):
        step(*inputs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(*inputs)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--only", default=None, help="substring of the case name (profiling runs)")
    ap.add_argument("--kinds", default="engine,aten")
    args = ap.parse_args()
    gen = torch.Generator().manual_seed(1)
    U, N, E = 22363, 12101, 153776                  # Beauty's graph size
    u, i = bipartite(U, N, E, gen)
    A_sym = sym_adj(U, N, u, i)
    A_left = sym_adj(U, N, u, i, self_loops=True, left=True)
    B = 2048
    users, pos, neg = (torch.randint(0, m, (B,), generator=gen).cuda() for m in (U, N, N))
    counts = [957, 4082, 7, 7, 2, 3, 2, 9, 80, 233]                                   # Frappe_x1
    x = torch.stack([torch.randint(0, c, (4096,), generator=gen) for c in counts], 1).cuda()
    y = (torch.rand(4096, generator=gen) < 0.3).float().cuda()
    S, Bs = 50, 512
    lens = torch.randint(1, S, (Bs,), generator=gen)
    seq_r = torch.zeros(Bs, S, dtype=torch.long); seq_l = torch.zeros(Bs, S, dtype=torch.long)
    for b in range(Bs):
        it = torch.randint(0, N, (int(lens[b]),), generator=gen)
        seq_r[b, : int(lens[b])] = it + 1
        seq_l[b, S - int(lens[b]):] = it + 2
    seq_r, seq_l = seq_r.cuda(), seq_l.cuda()
    seq_l1 = (seq_l - 1).clamp_min(0)               # left-padded, ids + 1 (STAMP / FMLP-Rec)
    p1, n1 = (torch.randint(0, N, (Bs,), generator=gen).cuda() for _ in range(2))

    tri = (users, pos, neg)
    cases = {   # name: (constructor, inputs of fit, capturable as a hipGraph (no host sync inside fit))
        "DCN (Frappe fields, B=4096)": (lambda: sib.DCN(counts, 10, (400, 400, 400), 3, batch_norm=True), (x, y), True),
        "SimGCL (Beauty graph, B=2048)": (lambda: sib.SimGCL(U, N, A_sym, 64, 3, eps=0.1), tri, True),
        "NGCF (Beauty graph, B=2048)": (lambda: sib.NGCF(U, N, A_left, 64, 3), tri, True),
        "JGCF (Beauty graph, B=2048)": (lambda: sib.JGCF(U, N, A_sym, 64, 3), tri, True),
        "GCN (Beauty graph, B=2048)": (lambda: sib.GCN(U, N, A_sym, 64, 3), tri, True),
        "STAMP BCE (N=12101, B=512, S=50)": (lambda: sib.STAMP(N, 64, 64, loss="BCE"), (seq_l1, p1, n1), True),
        "FMLP-Rec BPR (N=12101, B=512, S=50)": (lambda: sib.FMLPRec(N, S, 64, 2, loss="BPR"), (seq_l1, p1, n1), True),
        "NARM (N=12101, B=512, S=50)": (lambda: sib.NARM(N, 64, 128), (seq_r, p1, n1), "static"),
        "GRU4Rec BPR (N=12101, B=512, S=50)": (lambda: sib.GRU4Rec(N, 64, 128, loss="BPR"), (seq_r, p1, n1), "static"),
        "BERT4Rec (N=12101, B=512, S=50)": (lambda: sib.BERT4Rec(N, S, 64, 4, 2), (seq_l,), "static"),
    }
    out = {}
    for name, (make, inputs, graphable) in cases.items():
        if args.only and args.only not in name:
            continue
        row = {}
        for kind, table in (("engine", ENGINE), ("aten", ATEN)):
            if kind not in args.kinds.split(","):
                continue
            use(table)
            # (aten's operators are measured eagerly only: their captured DCN step ended in a GPU memory fault on this image)
            for mode in ("eager", "graph") if graphable and kind == "engine" else ("eager",):
                print(f"[{name}] {kind} {mode} ...", file=sys.stderr, flush=True)
                torch.manual_seed(3)
                m = make().cuda()
                m.train()
                if mode == "graph" and graphable == "static":
                    m.static_shapes = True          # (no shrink_pads / mask indexing: what the Coach sets for a captured step)
                try:
                    if mode == "eager":
                        row[f"{kind}_ms"] = round(timed(m, (lambda m=m: m.fit(*inputs)), args.steps), 4)
                    else:
                        row[f"{kind}_graph_ms"] = round(timed_graph(m, (lambda *a, m=m: sum(m.fit(*a).values())), inputs, args.steps), 4)
                except Exception as e:  # noqa: BLE001  (an aten operator without a ROCm backward or not capturable, e.g. CSR @ dense)
                    row[f"{kind}_{mode}_ms"] = f"{type(e).__name__}: {str(e)[:160]}"
                    torch.cuda.synchronize()
                del m
        use(ENGINE)
        out[name] = row
        print(name, row, file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
