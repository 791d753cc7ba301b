"""Two PROCESSES run the same captured SASRec steps (same seed, same batches, lr > 0); the parent reports the first step after which their
parameter arenas differ and in which tensors.     python scripts/soak_diff.py [--B 2048] [--steps 40] [--pipelined 0]"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--child", default="")
ap.add_argument("--B", type=int, default=2048)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--pipelined", type=int, default=0)
ap.add_argument("--lr", type=float, default=5e-4)
ap.add_argument("--tile", type=int, default=1)
a = ap.parse_args()
if a.child:
    import numpy as np, torch, bench
    from recboard_amd.sasrec import SASRecEngine
    cfg = dict(bench.BEAUTY, B=a.B)
    bs = [tuple(torch.from_numpy(x).cuda() for x in b) for b in bench.synth_batches(cfg, 8, 1)]
    m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=a.lr, weight_decay=1e-6, seed=1)
    if not a.tile:
        m.tile_step = False
    snaps = {}
    for i in range(a.steps):
        loss = m.train_step_graph(*bs[i % 8], next_batch=bs[(i + 1) % 8] if a.pipelined else None)
        W = m._bufs[(a.B, 50)]
        snaps[f"p{i}"] = m.arena.data.cpu().numpy().copy()
        snaps[f"g{i}"] = m.arena.grad.cpu().numpy().copy()
        snaps[f"l{i}"] = loss.cpu().numpy().copy()
        snaps[f"grows{i}"] = (W["g_rows"] * (W["keys"] != 0).unsqueeze(-1)).cpu().numpy().copy()
        snaps[f"keys{i}"] = W["keys"].cpu().numpy().copy()
        if i <= 2:      # the forward tape of both blocks, the rows of every tile
            NR = 16 * a.B * 4
            ACT = NR * 64
            per_block = 9 * ACT + NR * 64 + 3 * NR * 2 + NR * 16
            nt = int(m._graphs[(a.B, 50, True, bool(a.pipelined))]["blob"].numel() and m.prepare_batch(*bs[i % 8]).plan.view(torch.int32)[1]) if False else 1024
            for l in range(2):
                for j, n in enumerate(["X", "A", "Q", "K", "V", "O", "X1", "Y", "HR", "P"]):
                    snaps[f"tape{i}.{l}.{n}"] = W["tape"][l * per_block + j * ACT: l * per_block + j * ACT + nt * 16 * 64].cpu().numpy().copy().reshape(nt * 16, 64)
            snaps[f"u{i}"] = W["u"].reshape(-1, 64).cpu().numpy().copy()
            mt = a.B * 4
            wsf = W["ws_bwd"].view(torch.float32)
            o_gt = max(mt, 1024) * 2 * 12 * 64 + 2 * 6 * 24 * 64 * 64 + 64 * ((a.B + 63) // 64) * 64
            for l in range(2):
                for k_, n in enumerate(["dz", "dh", "dx1", "dq", "dk", "dv"]):
                    b0 = o_gt + (l * 6 + k_) * NR * 64
                    snaps[f"gtape{i}.{l}.{n}"] = wsf[b0:b0 + nt * 16 * 64].cpu().numpy().copy().reshape(nt * 16, 64)
            snaps[f"dU{i}"] = W["dU_rows"][:nt * 16].cpu().numpy().copy()
    m.check_handover()
    np.savez(a.child, **snaps)
    names = {k: (int(o), int(np.prod(s))) for (k, s), o in zip(m.arena.shapes.items(), m.arena.offsets.values())} if hasattr(m.arena, "offsets") else {}
    json.dump(names, open(a.child + ".json", "w"))
    sys.exit(0)
import numpy as np
outs = []
for r in range(2):
    f = f"/tmp/soak_diff_{r}.npz"
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", f, "--B", str(a.B), "--steps", str(a.steps), "--pipelined", str(a.pipelined),
                    "--lr", str(a.lr), "--tile", str(a.tile)], check=True)
    outs.append(np.load(f))
x, y = outs
for i in range(a.steps):
    d = {k: not np.array_equal(x[f"{k}{i}"], y[f"{k}{i}"]) for k in ("l", "keys", "grows", "g", "p")}
    if any(d.values()):
        print(f"first difference after step {i}: " + ", ".join(f"{k} {'DIFF' if v else 'same'}" for k, v in d.items()))
        for k in ("grows", "g", "p"):
            dd = np.nonzero(x[f"{k}{i}"].reshape(-1) != y[f"{k}{i}"].reshape(-1))[0]
            if dd.size:
                print(f"  {k}: {dd.size} elements differ, first at {dd[:6].tolist()}, last at {int(dd[-1])}, of {x[f'{k}{i}'].size}; max |diff| {float(np.abs(x[f'{k}{i}'].reshape(-1)[dd] - y[f'{k}{i}'].reshape(-1)[dd]).max()):.3e}")
        if i <= 2:
            for l in range(2):
                for n in ["X", "A", "Q", "K", "V", "O", "X1", "Y", "HR", "P"]:
                    k = f"tape{i}.{l}.{n}"
                    rows = np.nonzero((x[k] != y[k]).any(1))[0]
                    if rows.size:
                        r0 = int(rows[0])
                        cols = np.nonzero(x[k][r0] != y[k][r0])[0]
                        print(f"  {k}: rows {rows.size} (tiles {sorted(set((rows // 16).tolist()))[:8]}), first row {r0} cols {cols[:8].tolist()}.. n {cols.size}; "
                              f"x {x[k][r0][cols[:3]].tolist()} y {y[k][r0][cols[:3]].tolist()}")
            for l in (1, 0):
                for n in ["dz", "dh", "dx1", "dq", "dk", "dv"]:
                    k = f"gtape{i}.{l}.{n}"
                    rows = np.nonzero((x[k] != y[k]).any(1))[0]
                    if rows.size:
                        r0 = int(rows[0])
                        cols = np.nonzero(x[k][r0] != y[k][r0])[0]
                        print(f"  {k}: rows {rows.size} {rows[:16].tolist()} (tiles {sorted(set((rows // 16).tolist()))[:8]}), first row {r0}: cols n {cols.size} {cols[:10].tolist()}; "
                              f"x {x[k][r0][cols[:3]].tolist()} y {y[k][r0][cols[:3]].tolist()}")
            dr = np.nonzero((x[f"dU{i}"] != y[f"dU{i}"]).any(1))[0]
            print(f"  dU_rows: rows differing {dr.size}")
            ur = np.nonzero((x[f"u{i}"] != y[f"u{i}"]).any(1))[0]
            print(f"  u: rows differing {ur.size}")
        break
else:
    print(f"identical for {a.steps} steps (B {a.B}, pipelined {a.pipelined}, lr {a.lr})")
