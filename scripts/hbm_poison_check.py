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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="product")
    ap.add_argument("--B", type=int, default=2048)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--only", default="", help="comma list of buffer names to poison (default: all)")
    a = ap.parse_args()
    if a.lib != "product":
        lib.LIB_PATH = os.path.join(ROOT, "recboard_amd", f"librecengine_{a.lib}.so")
    lib.load()
    import bench
    from recboard_amd import ops
    from recboard_amd.sasrec import SASRecEngine
    cfg = dict(bench.BEAUTY, B=a.B)
    bs = [tuple(torch.from_numpy(x).cuda() for x in b) for b in bench.synth_batches(cfg, 2, 1)]
    m = SASRecEngine(cfg["items"], 50, a.dim, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, seed=1)
    A = m.arena
    print(f"lib {a.lib}: tile workgroups per CU {m._tile_wgs()}, B {a.B}, D {a.dim}", flush=True)
    init = [t.clone() for t in (A.data, A.m, A.v)]
    S = 50
    m.train_step_graph(*bs[0])                        # captures (the warm-up touches every workspace), runs once
    torch.cuda.synchronize()
    W = m._buffers(a.B, S)
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
