"""Does the captured SASRec step read HBM that nobody wrote?  (the one control the two-workgroups-per-CU notes lacked.)

ONE process, one engine, one batch: the step is run with every workspace that the allocation contract leaves UNINITIALISED (`torch.empty`:
u, dU, contrib, dU_rows, g_rows, the backward workspace = slabs / partials / gradient tape / weight fragments / dK-dV inboxes, the scatter
workspace) and the activation tape outside its hand-over flag words pre-filled with a pattern -- 0x00, 0xFF (NaN), 1.0f, a large negative -- and
the parameters, moments and step count put back in between.  If any output bit depends on the pattern, some launch reads memory no launch of
the step wrote (which would also explain results that are stable inside a process and differ between processes); the script then bisects by
buffer.  Run on the product library (one workgroup per CU: the control) and on `make two` (--lib two).

    python scripts/hbm_poison_check.py --lib two --B 2048"""
import argparse
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from recboard_amd import lib  # noqa: E402

PATTERNS = {"zero": 0x00000000, "nan": 0xFFFFFFFF, "one": 0x3F800000, "neg": 0xFF61B1E6}


def fill_words(t, word, lo=0, hi=None):
    v = t.reshape(-1).view(torch.uint8)
    n = v.numel() // 4 * 4
    w = v[:n].view(torch.int32)
    w[lo:hi] = word if word < 0x80000000 else word - (1 << 32)


def locate(a):
    """Two engines in ONE process on an experiment library whose tile launch takes its LDS request from RE_TILE_LDS_KB (-DTL_LDS_KB_ENV; the
    value is frozen into the captured graph): the reference engine recorded at 84 KB (one workgroup per CU: deterministic), the other at 58 KB
    (two per CU).  After every repetition of the same step (lr = 0) every intermediate buffer of the second engine is compared with the
    reference's, and the differing words are decoded into (array, block, tile, row, column)."""
    lib.LIB_PATH = os.path.join(ROOT, "recboard_amd", f"librecengine_{a.lib}.so")
    lib.load()
    import collections
    import bench
    from recboard_amd import ops
    from recboard_amd.sasrec import SASRecEngine
    cfg = dict(bench.BEAUTY, B=a.B)
    bs = [tuple(torch.from_numpy(x).cuda() for x in b) for b in bench.synth_batches(cfg, 2, 1)]
    S, D, L = 50, a.dim, 2

    def engine(kb):
        os.environ["RE_TILE_LDS_KB"] = str(kb)
        m = SASRecEngine(cfg["items"], S, D, L, dropout_rate=0.5, lr=0.0, weight_decay=0.0, seed=1)
        m.train_step_graph(*bs[0])
        torch.cuda.synchronize()
        return m
    ref, two = engine(84), engine(58)
    NR = ops.sasrec_plan_rows(a.B, S)
    act = NR * D
    names_blk = ["X", "A", "Q", "K", "V", "O", "X1", "Y", "HR"]
    per_block = 9 * act + NR * 64 + 3 * NR * 2 + NR * (D // 4)
    gnames = ["dO2", "dH", "dX1", "dQ", "dK", "dV"]
    slab_f = max(a.B * 4, 1024) * L * 12 * D
    wpart_f = L * 6 * 24 * D * D
    ppart_f = 64 * ((a.B + 63) // 64) * D
    g0 = slab_f + wpart_f + ppart_f

    def where_tape(i):
        if i >= L * per_block:
            j = i - L * per_block
            return ("XL", -1, j // D // 16, j // D % 16, j % D) if j < act else ("SL/flags", -1, -1, -1, j - act)
        l, o = divmod(i, per_block)
        if o < 9 * act:
            k, j = divmod(o, act)
            return (names_blk[k], l, j // D // 16, j // D % 16, j % D)
        o -= 9 * act
        if o < NR * 64:
            return ("P", l, o // 64 // 16, o // 64 % 16, o % 64)
        return ("SA/SF/PP/MK", l, -1, -1, o - NR * 64)

    def where_ws(i):
        if i < slab_f:
            return ("slab", -1, i // (L * 12 * D), -1, i % (12 * D))
        if i < slab_f + wpart_f:
            return ("wpart", -1, -1, -1, i - slab_f)
        if i < g0:
            return ("ppart", -1, -1, -1, i - slab_f - wpart_f)
        j = i - g0
        if j < L * 6 * act:
            l, o = divmod(j, 6 * act)
            k, o = divmod(o, act)
            return (gnames[k], l, o // D // 16, o // D % 16, o % D)
        return ("wf/xch", -1, -1, -1, j - L * 6 * act)

    def snap(m):
        W = m._buffers(a.B, S)
        out = {k: W[k].reshape(-1).view(torch.uint8)[: W[k].numel() * W[k].element_size() // 4 * 4].view(torch.int32).clone() for k in ("u", "dU_rows", "g_rows", "keys", "tape", "ws_bwd")}
        out["grad"] = m.arena.grad.view(torch.int32).clone()
        return out

    def step(m):
        m.arena.step = 0
        m.train_step_graph(*bs[0])
        torch.cuda.synchronize()
        return snap(m)
    R = step(ref)
    R2 = step(ref)
    print("reference engine (84 KB) repeats itself:", all(torch.equal(R[k], R2[k]) for k in R if k not in ("tape", "ws_bwd")) and
          torch.equal(R["tape"][: L * per_block + act], R2["tape"][: L * per_block + act]) and torch.equal(R["ws_bwd"][g0: g0 + L * 6 * act], R2["ws_bwd"][g0: g0 + L * 6 * act]), flush=True)
    hist = collections.Counter()
    for rep in range(a.reps):
        T = step(two)
        line = []
        for k in ("tape", "u", "dU_rows", "g_rows", "keys", "ws_bwd", "grad"):
            d = torch.nonzero(T[k] != R[k]).reshape(-1)
            if k == "ws_bwd":       # (the gradient tape only: slabs / partials are sums over it, the inboxes and scratch behind it are not results)
                d = d[(d >= g0) & (d < g0 + L * 6 * act)]
            if k == "tape":         # (not the hand-over flag / epoch words at the end)
                d = d[d < L * per_block + act]
            if d.numel() == 0:
                continue
            idx = d[:200000].cpu().tolist()
            if k == "ws_bwd":
                per = collections.defaultdict(lambda: [0, set(), set(), set()])
                for i in idx:
                    n, l, t, r, c = where_ws(i)
                    e = per[(n, l)]
                    e[0] += 1; e[1].add(t); e[2].add(c); e[3].add(r)
                # the FIRST wrong array of the backward (dO2 of the highest block that differs): every wrong word beside the reference's, and
                # whether its bit pattern exists anywhere in the reference's buffers (a copy of another element?) 
                first = max((l for (n, l) in per if n == "dO2"), default=None)
                if first is not None:
                    sel = [i for i in idx if where_ws(i)[:2] == ("dO2", first)][:24]
                    allref = torch.cat([R[q] for q in ("tape", "ws_bwd", "g_rows", "dU_rows", "u")])
                    if first == L - 1:
                        # site A (LN_last backward): dx = rstd (du g - s1 - xh s2), dO2 = 2 dx where dropout kept.  Which ONE input, replaced by
                        # what, gives the wrong word?  (reference du = dU_rows, x_L / (mean, rstd) from the tape, g = lastLN.weight)
                        tf = R["tape"].view(torch.float32)
                        XL = tf[L * per_block: L * per_block + act].view(NR, D).double()
                        SLs = tf[L * per_block + act: L * per_block + act + NR * 2].view(NR, 2).double()
                        DU = R["dU_rows"].view(torch.float32).view(NR, D).double()
                        gam = ref.params["lastLN.weight"].double()
                        for i in sel[:12]:
                            n, l, t, r, c = where_ws(i)
                            row = t * 16 + r
                            mean, rstd = SLs[row, 0], SLs[row, 1]
                            xh = (XL[row] - mean) * rstd
                            dd = DU[row] * gam
                            s1, s2 = dd.mean(), (dd * xh).mean()
                            want = rstd * (dd[c] - s1 - xh[c] * s2)
                            got = T[k].view(torch.float32)[i].double() / 2.0
                            refv = R[k].view(torch.float32)[i].double() / 2.0
                            d_alt = got / rstd + s1 + xh[c] * s2                 # the d[c] that would explain it
                            x_alt = (dd[c] - s1 - got / rstd) / s2              # the xh[c] that would explain it
                            near_d = torch.argmin((dd - d_alt).abs()).item(); near_x = torch.argmin((xh - x_alt).abs()).item()
                            print(f"    A: tile {t} row {r} col {c}: host {want.item():.6e} ref {refv.item():.6e} got {got.item():.6e} | d[c] {dd[c].item():.4e} would need {d_alt.item():.4e} "
                                  f"(nearest d in the row: col {near_d} {dd[near_d].item():.4e}) | xh[c] {xh[c].item():.4f} would need {x_alt.item():.4f} (nearest xh: col {near_x} {xh[near_x].item():.4f}); s1 {s1.item():.3e} s2 {s2.item():.3e}", flush=True)
                    for i in sel:
                        n, l, t, r, c = where_ws(i)
                        wv, rv = T[k][i].item(), R[k][i].item()
                        hits = torch.nonzero(allref == wv).reshape(-1)[:4].cpu().tolist()
                        fw, fr_ = T[k].view(torch.float32)[i].item(), R[k].view(torch.float32)[i].item()
                        # the same token's other dO2 columns of this strip (reference), for scale
                        print(f"    dO2[{l}] tile {t} row {r} col {c}: got {fw:.9e} want {fr_:.9e} ratio {fw / fr_ if fr_ else float('nan'):.6f}; bit pattern elsewhere in the reference at {hits}", flush=True)
                order = ["dO2", "dH", "dX1", "dQ", "dK", "dV"]
                line.append("gtape: " + "; ".join(f"{n}[{l}] {e[0]} words tiles {sorted(e[1])[:4]} cols {sorted(e[2])[:20]} rows {sorted(e[3])}"
                                                    for (n, l), e in sorted(per.items(), key=lambda kv: (-kv[0][1], order.index(kv[0][0])))))
                continue
            if k == "tape":
                locs = [where_tape(i) for i in idx]
            elif k == "ws_bwd":
                locs = [where_ws(i) for i in idx]
            elif k in ("u", "dU_rows"):
                locs = [(k, -1, i // D // 16, i // D % 16, i % D) for i in idx]
            elif k == "g_rows":
                locs = [(f"g_rows[{i // act}]", -1, i % act // D // 16, i % act // D % 16, i % D) for i in idx]
            else:
                locs = [(k, -1, -1, -1, i) for i in idx]
            groups = collections.Counter((n, l, t, c) for n, l, t, r, c in locs)
            for (n, l, t, c), cnt in list(groups.items())[:6]:
                hist[(n, l, c % 16 if c >= 0 and n not in ("grad", "keys") else -1)] += 1
            fa = T[k].view(torch.float32)[d[:3]].cpu().tolist(); fr = R[k].view(torch.float32)[d[:3]].cpu().tolist()
            tiles = sorted({t for n, l, t, r, c in locs})
            line.append(f"{k}: {d.numel()} words, tiles {tiles[:6]}, cols {sorted({c for n, l, t, r, c in locs})[:24]}")
        print(f"rep {rep}: " + ("identical" if not line else " | ".join(line)), flush=True)
    print("histogram (array, block, column mod 16):", sorted(hist.items(), key=lambda kv: -kv[1])[:40], flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="product")
    ap.add_argument("--B", type=int, default=2048)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--only", default="", help="comma list of buffer names to poison (default: all)")
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--item", action="store_true", help="the workgroup-per-item kernels (exact fp32: enc_step_k) instead of the tile kernels")
    ap.add_argument("--mode", default="poison", choices=("poison", "flush", "locate"),
                    help="flush: lr = 0, nothing reset or refilled between repetitions; what runs BETWEEN two repetitions varies instead -- nothing, a "
                         "one-word kernel, a 1 GiB fill (every cache line of the step's buffers evicted), a host synchronisation")
    a = ap.parse_args()
    if a.mode == "locate":
        return locate(a)
    if a.lib != "product":
        lib.LIB_PATH = os.path.join(ROOT, "recboard_amd", f"librecengine_{a.lib}.so")
    lib.load()
    import bench
    from recboard_amd import ops
    from recboard_amd.sasrec import SASRecEngine
    cfg = dict(bench.BEAUTY, B=a.B)
    bs = [tuple(torch.from_numpy(x).cuda() for x in b) for b in bench.synth_batches(cfg, 2, 1)]
    m = SASRecEngine(cfg["items"], 50, a.dim, 2, dropout_rate=0.5, lr=0.0 if a.mode == "flush" else 5e-4, weight_decay=0.0 if a.mode == "flush" else 1e-6, seed=1)
    if a.item:
        m.tile_step = False
    A = m.arena
    print(f"lib {a.lib}: tile workgroups per CU {m._tile_wgs()}, B {a.B}, D {a.dim}, kernels: {'workgroup per item' if a.item else 'tile'}", flush=True)
    init = [t.clone() for t in (A.data, A.m, A.v)]
    S = 50
    m.train_step_graph(*bs[0])                        # captures (the warm-up touches every workspace), runs once
    torch.cuda.synchronize()
    W = m._buffers(a.B, S)
    if a.mode == "flush":
        big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        one = torch.zeros(1, device="cuda")

        def rep(between, n=6):
            hs = []
            for i in range(n):
                A.step = 0                                   # (the same dropout seed every time)
                loss = m.train_step_graph(*bs[0])
                h = hashlib.sha1(loss.cpu().numpy().tobytes() + A.grad.cpu().numpy().tobytes()).hexdigest()[:12]
                hs.append(h)
                if between == "word":
                    one.add_(1.0)
                elif between == "fill":
                    big.fill_(float(i))
                elif between == "sync":
                    torch.cuda.synchronize()
                elif between == "sleep":
                    torch.cuda.synchronize(); __import__("time").sleep(0.2)
            return hs
        for between in ("none", "word", "sync", "sleep", "fill", "none"):
            hs = rep(between)
            print(f"between = {between:5s}: {len(set(hs))} distinct of {len(hs)}: {hs}", flush=True)
        return
    NR = ops.sasrec_plan_rows(a.B, S)
    flag_words = NR // 16 * 8 + 16                     # csrc/enc_common.h enc_tape_layout: the tape's tail (zero by contract)
    names = ["u", "dU", "contrib", "dU_rows", "g_rows", "ws_bwd", "ws_sc", "tape"]
    only = [s for s in a.only.split(",") if s] or names

    def run(word, which):
        for t, k in zip((A.data, A.m, A.v), init):
            t.copy_(k)
        A.grad.zero_()
        A.step = 0
        for nme in which:
            t = W[nme]
            if nme == "tape":
                fill_words(t, word, 0, t.numel() - flag_words)
            else:
                fill_words(t, word)
        torch.cuda.synchronize()
        h = hashlib.sha1()
        for i in range(a.steps):
            loss = m.train_step_graph(*bs[i % 2])
            torch.cuda.synchronize()
            h.update(loss.cpu().numpy().tobytes()); h.update(A.grad.cpu().numpy().tobytes()); h.update(A.data.cpu().numpy().tobytes())
        m.check_handover()
        return h.hexdigest()[:16]

    base = {p: run(w, only) for p, w in PATTERNS.items()}
    print("all buffers:", base, flush=True)
    rep = run(PATTERNS["zero"], only)
    print("repeat of `zero`:", rep, "(same as the first)" if rep == base["zero"] else "(DIFFERS from the first run with the same fill: not a fill effect)", flush=True)
    if len(set(base.values())) == 1:
        print("RESULT: every pattern gives the same bits -- the step reads no HBM workspace word it did not write", flush=True)
        return
    for nme in only:
        r = {p: run(w, [nme]) for p, w in PATTERNS.items()}
        print(f"only {nme}:", r, "<-- depends on the fill" if len(set(r.values())) > 1 else "", flush=True)


if __name__ == "__main__":
    main()
