"""The SAME SASRec training step (same parameters, same batch, same dropout seed, lr = 0) over and over in one process, on the hand-over
diagnostic build: which gradient tensors ever differ from the first repetition's, by how much, how often.
    python scripts/handover_repeat.py --lds-kb 60 --reps 2000"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="hovn")
ap.add_argument("--fenced", type=int, default=0)
ap.add_argument("--lds-kb", type=int, default=60)
ap.add_argument("--reps", type=int, default=2000)
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--graph", type=int, default=1)
ap.add_argument("--cycle", type=int, default=1, help="cycle through this many different batches (each compared with its own first occurrence)")
ap.add_argument("--pipelined", type=int, default=0)
ap.add_argument("--paranoid", type=int, default=0)
ap.add_argument("--sync", type=int, default=0, help="1: torch.cuda.synchronize() in front of every step")
ap.add_argument("--buffers", type=int, default=0, help="1: also compare the tape / contribution-row buffers of every repetition")
ap.add_argument("--B", type=int, default=0, help="sequences per batch (default: the bench's 512; more than 512: the looped form of the tile kernel)")
a = ap.parse_args()
import torch  # noqa: E402

from recboard_amd import lib  # noqa: E402
lib.LIB_PATH = os.path.join(ROOT, "recboard_amd", f"librecengine_{a.lib}.so")
L = lib.load()
L.re_dbg_tile_handover.argtypes, L.re_dbg_tile_handover.restype = [ctypes.c_int, ctypes.c_int], ctypes.c_int
assert L.re_dbg_tile_handover(a.fenced, a.lds_kb) == 0
if a.paranoid:
    L.re_dbg_tile_paranoid.argtypes, L.re_dbg_tile_paranoid.restype = [ctypes.c_uint], ctypes.c_int
    assert L.re_dbg_tile_paranoid(a.paranoid) == 0
import bench  # noqa: E402
from recboard_amd.sasrec import SASRecEngine  # noqa: E402
cfg = dict(bench.BEAUTY, B=a.B) if a.B else bench.BEAUTY
bs = [tuple(torch.from_numpy(x).cuda() for x in b) for b in bench.synth_batches(cfg, 8, 1)]
m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=0.0, weight_decay=0.0, seed=1)
A = m.arena
refs = {}
tiles = {j: int(m.prepare_batch(*bs[j]).plan.view(torch.int32)[1]) for j in range(a.cycle)}
Bsz, S = bs[0][0].shape
NR = 16 * Bsz * ((S + 15) // 16)
ACT = NR * 64
names = ["X", "A", "Q", "K", "V", "O", "X1", "Y", "HR"]
per_block = 9 * ACT + NR * 64 + 3 * NR * 2 + NR * 16


def snapshot(j):
    W = m._bufs[(Bsz, S)]
    nr = 16 * tiles[j]
    out = {}
    for l in range(2):
        base = l * per_block
        for i, n in enumerate(names):
            out[f"tape.{l}.{n}"] = W["tape"][base + i * ACT: base + i * ACT + nr * 64].clone().view(nr, 64)
        out[f"tape.{l}.P"] = W["tape"][base + 9 * ACT: base + 9 * ACT + nr * 64].clone().view(nr, 64)
    for r in range(3):
        out[f"g_rows.{r}"] = W["g_rows"][r, :nr].clone() * (W["keys"][r, :nr] != 0).unsqueeze(1)     # (rows without a key are not consumed: whatever an earlier step left)
        out[f"keys.{r}"] = W["keys"][r, :nr].clone().unsqueeze(1)
    out["dU_rows"] = W["dU_rows"][:nr].clone()
    # the gradient tape and the per-tile slab in the backward workspace (csrc/enc_step.hip: slab | wpart | ppart | gtape)
    mt = Bsz * ((S + 15) // 16)
    wsf = W["ws_bwd"].view(torch.float32)
    slab_rows = max(mt, 1024)
    o_slab = 0
    o_gt = slab_rows * 2 * 12 * 64 + 2 * 6 * 24 * 64 * 64 + 64 * ((Bsz + 63) // 64) * 64
    for l in range(2):
        for k, n in enumerate(["dz", "dh", "dx1", "dq", "dk", "dv"]):
            b0 = o_gt + (l * 6 + k) * NR * 64
            out[f"gtape.{l}.{n}"] = wsf[b0:b0 + nr * 64].clone().view(nr, 64)
    out["slab"] = wsf[:tiles[j] * 2 * 12 * 64].clone().view(tiles[j], 2 * 12 * 64)
    out["u"] = W["u"].reshape(-1, 64).clone()
    return out


bad = {}
badbuf = {}
detail = []
nbad = 0
for r in range(a.reps):
    A.step = 0
    if a.sync:
        torch.cuda.synchronize()
    j = (a.batch + r) % a.cycle
    b = bs[j]
    if a.graph:
        loss = m.train_step_graph(*b, next_batch=bs[(a.batch + r + 1) % a.cycle] if a.pipelined else None)
    else:
        loss = m.train_step(*b)
    g = A.grad.clone()
    snap = snapshot(j) if a.buffers else None
    if j not in refs:
        refs[j] = (g, loss.clone(), snap)
        continue
    ref, refloss, rsnap = refs[j]
    if snap is not None and not torch.equal(g, ref):
        desc = []
        for k, v in snap.items():
            d = (v != rsnap[k]).any(1).nonzero().reshape(-1)
            if d.numel():
                desc.append(f"{k}[tiles {sorted(set((d // (1 if k == 'slab' else 16)).tolist()))[:6]} rows {d.numel()}]")
        badbuf[r] = (j, tiles[j], desc)
        if len(detail) < 3:
            W = m._bufs[(Bsz, S)]
            full_new, full_ref = W["g_rows"].clone(), None
            k = "g_rows.1"
            d = snap[k] - rsnap[k]
            rows = (d != 0).any(1).nonzero().reshape(-1)
            if rows.numel():
                r0 = int(rows[0])
                cols = (d[r0] != 0).nonzero().reshape(-1)
                c0 = int(cols[0]) // 16 * 16
                detail.append(f"rep {r} g_rows.1 row {r0} cols {c0}..{c0 + 15}: new {[float('%.3e' % x) for x in snap[k][r0, c0:c0 + 16].tolist()]}")
                detail.append(f"rep {r} g_rows.1 row {r0} cols {c0}..{c0 + 15}: ref {[float('%.3e' % x) for x in rsnap[k][r0, c0:c0 + 16].tolist()]}")
                # does the reference strip of this row show up anywhere else in the three regions (a misdirected store)?
                want = rsnap[k][r0, c0:c0 + 4]
                G3 = full_new.reshape(-1, 4)
                hit = ((G3 - want).abs().max(1).values == 0).nonzero().reshape(-1)
                detail.append(f"rep {r}: the reference strip's first 4 values found at float offsets {[int(h) * 4 for h in hit[:6].tolist()]} (expected {(NR + r0) * 64 + c0})")
                detail.append(f"rep {r}: keys.1 rows {snap['keys.1'][rows[:4], 0].tolist()} ref {rsnap['keys.1'][rows[:4], 0].tolist()}; raw new row (unmasked) {[float('%.3e' % x) for x in full_new[1, r0, c0:c0 + 4].tolist()]}")
        for k in ("gtape.1.dz", "gtape.1.dx1", "gtape.0.dz", "g_rows.0", "g_rows.1", "g_rows.2", "keys.1", "dU_rows"):
            d = (snap[k] - rsnap[k])
            rows = (d != 0).any(1).nonzero().reshape(-1)
            if rows.numel() and len(detail) < 6:
                r0 = int(rows[0])
                strips = [float(d[rows].abs().float()[:, 16 * q:16 * q + 16].max()) if d.shape[1] == 64 else 0.0 for q in range(4)]
                ratio = (snap[k][r0].float() / rsnap[k][r0].float())
                detail.append(f"rep {r} {k}: rows {rows.tolist()[:8]} per-strip max|diff| {['%.2e' % x for x in strips]} max|ref| {float(rsnap[k][rows].abs().float().max()):.2e} "
                              f"ratio new/ref of row {r0}: min {float(ratio.min()):.4f} max {float(ratio.max()):.4f}")
    if not torch.equal(g, ref) or not torch.equal(loss, refloss):
        nbad += 1
        for k, v in A.views(g).items():
            d = (v - A.views(ref)[k]).abs()
            if float(d.max()) > 0:
                rows = int((d.reshape(d.shape[0], -1).max(1).values > 0).sum()) if d.dim() > 1 else int((d > 0).sum())
                e = bad.setdefault(k, [0, 0.0, 0])
                e[0] += 1; e[1] = max(e[1], float(d.max() / (A.views(ref)[k].abs().max() + 1e-30))); e[2] = max(e[2], rows)
torch.cuda.synchronize()
if a.lib in ("hov", "hovn"):
    L.re_dbg_tile_stale.argtypes, L.re_dbg_tile_stale.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
    buf = (ctypes.c_uint * 8)()
    assert L.re_dbg_tile_stale(buf, 1) == 0
    print("checksum mismatches [forward k/v, backward k/v, inbox]:", list(buf[:3]), "of checks", list(buf[4:7]),
          "| LDS canary words changed:", buf[3], "| LDS parameter words changed:", buf[7])
for r, (j, nt, desc) in list(badbuf.items())[:12]:
    print(f"rep {r} batch {j} ({nt} tiles): " + "; ".join(desc))
for d in detail:
    print(d)
print(f"sync {a.sync} paranoid {a.paranoid} lds_kb {a.lds_kb} fenced {a.fenced} graph {a.graph} cycle {a.cycle} pipelined {a.pipelined}: {nbad} of {a.reps - 1} repetitions differ from the first")
for k, (n, rel, rows) in sorted(bad.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"  {k:40s} differs in {n:5d} reps; max |diff| / max |grad| {rel:.2e}; rows (or elements) touched <= {rows}")
