"""Hand-over soak of the tile kernel (csrc/enc_tile.hip): N PROCESS pairs x STEPS captured SASRec steps at Beauty's shapes, per configuration
{protocol: sc1-only | agent release / acquire} x {workgroups per CU: 1 (84 KB of LDS requested) | 2 (60 KB)}, on the hand-over diagnostic
build (`make -C recboard_amd/csrc hov`; --lib hovn: the same switches WITHOUT the checksums and their extra barriers): every block of rows that crosses workgroups travels with a checksum of its bit patterns and the
consumer checks what it LOADED against it.  Reported per configuration: checks made, checksum mismatches (a stale or torn row OBSERVED),
and how many process pairs ended with different parameters.

    python scripts/handover_soak.py --pairs 20 --steps 300            # the driver: children run one after another
    python scripts/handover_soak.py --child --fenced 0 --lds-kb 60    # one process: prints one JSON line"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    import ctypes

    import torch

    from recboard_amd import lib
    if a.lib != "product":
        lib.LIB_PATH = os.path.join(ROOT, "recboard_amd", f"librecengine_{a.lib}.so")
    L = lib.load()
    if a.lib in ("hov", "hovn", "hovs"):
        L.re_dbg_tile_handover.argtypes, L.re_dbg_tile_handover.restype = [ctypes.c_int, ctypes.c_int], ctypes.c_int
        L.re_dbg_tile_stale.argtypes, L.re_dbg_tile_stale.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
        assert L.re_dbg_tile_handover(a.fenced, a.lds_kb) == 0
        if a.fill:
            L.re_dbg_tile_fill.argtypes, L.re_dbg_tile_fill.restype = [ctypes.c_uint], ctypes.c_int
            assert L.re_dbg_tile_fill(int(a.fill, 0)) == 0
    import bench
    from recboard_amd.sasrec import SASRecEngine
    cfg = dict(bench.BEAUTY, B=a.B) if a.B else bench.BEAUTY
    bs = [tuple(torch.from_numpy(x).cuda() for x in b) for b in bench.synth_batches(cfg, 8, 1)]
    m = SASRecEngine(cfg["items"], 50, a.dim, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
    for i in range(a.steps):
        m.train_step_graph(*bs[i % 8], next_batch=bs[(i + 1) % 8] if a.pipelined else None)
    torch.cuda.synchronize()
    m.check_handover()
    out = {"sha1": hashlib.sha1(m.arena.data.cpu().numpy().tobytes()).hexdigest()[:16]}
    if a.lib == "hov":
        buf = (ctypes.c_uint * 8)()
        assert L.re_dbg_tile_stale(buf, 1) == 0
        out["stale"], out["checks"] = list(buf[:3]), list(buf[4:7])
    print("SOAK " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--lib", default="hov", choices=("hov", "hovn", "hovs", "fenced", "product", "ss", "two", "twoinv"))
    ap.add_argument("--fenced", type=int, default=0)
    ap.add_argument("--lds-kb", type=int, default=84)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--pairs", type=int, default=20)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--pipelined", type=int, default=1)
    ap.add_argument("--fill", default="", help="bit pattern every workgroup fills its LDS with first (0x7fc00000 = NaN), diagnostic builds")
    ap.add_argument("--configs", default="0:60,0:84,1:60,1:84")
    ap.add_argument("--B", type=int, default=0, help="sequences per batch (default: the bench's 512; more: the looped form of the tile kernel)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.child:
        return child(a)
    report = {}
    for c in a.configs.split(","):
        fenced, kb = (int(x) for x in c.split(":"))
        runs = []
        for i in range(2 * a.pairs):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--lib", a.lib, "--fenced", str(fenced), "--lds-kb", str(kb),
                                "--steps", str(a.steps), "--dim", str(a.dim), "--pipelined", str(a.pipelined), "--B", str(a.B)] + (["--fill", a.fill] if a.fill else []), capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("SOAK ")]
            if r.returncode != 0 or not line:
                print(r.stdout[-2000:], r.stderr[-2000:])
                raise SystemExit(f"child failed in configuration {c}")
            runs.append(json.loads(line[0][5:]))
        pairs_diff = sum(runs[2 * i]["sha1"] != runs[2 * i + 1]["sha1"] for i in range(a.pairs))
        rec = {"protocol": "release/acquire" if fenced else "sc1 only", "workgroups_per_cu": 1 if kb > 80 else 2, "lds_kb": kb,
               "process_pairs": a.pairs, "steps": a.steps, "pairs_with_different_parameters": pairs_diff,
               "distinct_end_states": len({r["sha1"] for r in runs}), "end_states": sorted({r["sha1"] for r in runs})[:4]}
        if "stale" in runs[0]:
            rec["checks"] = [sum(r["checks"][k] for r in runs) for k in range(3)]
            rec["checksum_mismatches"] = [sum(r["stale"][k] for r in runs) for k in range(3)]
        report[c] = rec
        print(json.dumps({c: rec}), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"what": "scripts/handover_soak.py: [forward k / v, backward k / v, dK / dV inbox] checks and checksum mismatches summed over all processes",
                       "lib": a.lib, "dim": a.dim, "B": a.B or 512, "configs": report}, f, indent=1)


if __name__ == "__main__":
    main()
