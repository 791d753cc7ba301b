"""A/B of score_topk's two paths on one MI355X: the split form (bf16 hi/mid on the XDL pipe + exact re-scoring + certificate,
the default) against the exact fp32-MFMA kernel alone (re_dbg_score_x2(0)).  Results must be identical bit for bit; prints
the number of users the certificate sent to the fallback, the largest observed |s' - s| / eps, and both timings.
Usage: python scripts/x2_check.py [quick]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recboard_amd import lib, ops  # noqa: E402
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "librecengine_dbg.so")   # the re_dbg_* switches live in the diagnostic twin (make -C recboard_amd/csrc dbg)

L = lib.load()
L.re_dbg_score_x2.argtypes = [ctypes.c_int]; L.re_dbg_score_x2.restype = None
L.re_dbg_score_x2_maxerr.argtypes = [ctypes.c_int]; L.re_dbg_score_x2_maxerr.restype = None
L.re_dbg_score_x2_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.re_dbg_score_x2_stats.restype = None
L.re_dbg_score_x2_info.argtypes = [ctypes.c_void_p]; L.re_dbg_score_x2_info.restype = None


def stats(reset=True):
    torch.cuda.synchronize()
    out = (ctypes.c_uint32 * 2)()
    L.re_dbg_score_x2_stats(out, 1 if reset else 0)
    return int(out[0]), float(torch.tensor([out[1]], dtype=torch.int64).to(torch.int32).view(torch.float32)[0])


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it


def case(name, q, E, sp, si, K, time_it=True, prep=False):
    L.re_dbg_score_x2(0)
    v0, i0 = ops.score_topk(q, E, sp, si, K)
    t0 = timeit(lambda: ops.score_topk(q, E, sp, si, K)) if time_it else float("nan")
    L.re_dbg_score_x2(1)
    stats()
    L.re_dbg_score_x2_maxerr(1)
    pr = ops.score_prepare(E) if prep else None
    v1, i1 = ops.score_topk(q, E, sp, si, K, prep=pr)
    flagged, ratio = stats()
    if flagged:
        info = (ctypes.c_float * 8)()
        L.re_dbg_score_x2_info(info)
        print("    last flagged: user %d  T %.6g  x_K %.6g  eps %.3g  entries %d  lists %d  validK %d" % tuple(
            [int(info[0])] + [info[i] for i in (1, 2, 3)] + [int(info[4]), int(info[5]), int(info[6])]))
    L.re_dbg_score_x2_maxerr(0)
    t1 = timeit(lambda: ops.score_topk(q, E, sp, si, K, prep=pr)) if time_it else float("nan")
    ok = torch.equal(i0, i1) and torch.equal(v0.view(torch.int32), v1.view(torch.int32))
    nbad = int((i0 != i1).sum())
    B, N = q.shape[0], E.shape[0]
    tf = 2.0 * q.shape[1] * B * N / (t1 * 1e-3) / 1e12 if time_it else float("nan")
    print(f"{name:34s} B={B:6d} N={N:8d} D={q.shape[1]:3d} K={K:2d}  identical={ok} (idx diffs {nbad})  fallback users={flagged:5d}  "
          f"max err/eps={ratio:.4f}  exact {t0:.3f} ms  split {t1:.3f} ms  ({tf:.1f} TF)", flush=True)
    return ok


def seen(U, N, n, g):
    sp = torch.arange(0, U + 1, device="cuda") * n
    si = torch.sort(torch.randint(0, N, (U, n), device="cuda", generator=g), 1).values.reshape(-1)
    return sp, si


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    g = torch.Generator(device="cuda").manual_seed(1)
    ok = True
    U, N, D = 22363, 12101, 64
    q = torch.randn(U, D, device="cuda", generator=g); E = torch.randn(N, D, device="cuda", generator=g)
    sp, si = seen(U, N, 8, g)
    ok &= case("beauty iid", q, E, sp, si, 50)
    ok &= case("beauty iid, prepared", q, E, sp, si, 50, prep=True)
    ok &= case("beauty iid retain_seen", q, E, None, None, 50)
    for K in (1, 5, 10, 20, 26, 27, 44, 51):
        ok &= case(f"beauty iid K={K}", q, E, sp, si, K, time_it=False)
    # trained-like: popularity-scaled item norms, users = noisy mixtures of few items (clustered scores)
    En = E * (0.2 + 3.0 * torch.rand(N, 1, device="cuda", generator=g) ** 4)
    qc = En[torch.randint(0, N, (U,), device="cuda", generator=g)] + 0.3 * torch.randn(U, D, device="cuda", generator=g)
    ok &= case("clustered, wide norms", qc, En, sp, si, 50)
    # MF-BPR style tiny tables
    ok &= case("tables ~1e-4", q * 1e-4, E * 1e-4, sp, si, 50, time_it=False)
    # planted ties: every row appears 8 times; zero queries; identical queries
    Et = E[:N // 8].repeat(8, 1)[torch.randperm((N // 8) * 8, device="cuda", generator=g)]
    ok &= case("8-fold duplicate rows", q, Et.contiguous(), None, None, 50, time_it=False)
    qz = q.clone(); qz[::7] = 0.0
    ok &= case("every 7th query zero", qz, E, sp, si, 50, time_it=False)
    # quantised values: many exact ties between different rows
    Eq = torch.round(E * 2) / 2; qq = torch.round(q * 2) / 2
    ok &= case("half-integer tables (mass ties)", qq, Eq, sp, si, 50, time_it=False)
    # cancellation-heavy: large components that cancel
    Ec = E.clone(); Ec[:, :32] = 100.0 * Ec[:, :32]; qcn = q.clone(); qcn[:, 32:] = 100.0 * qcn[:, 32:]
    ok &= case("ill-scaled halves", qcn, Ec, sp, si, 50, time_it=False)
    if not quick:
        U2, N2 = 77277, 45638
        q2 = torch.randn(U2, D, device="cuda", generator=g); E2 = torch.randn(N2, D, device="cuda", generator=g)
        sp2, si2 = seen(U2, N2, 25, g)
        ok &= case("yelp iid", q2, E2, sp2, si2, 50)
        q3 = torch.randn(512, D, device="cuda", generator=g); E3 = torch.randn(12_500_000, D, device="cuda", generator=g)
        ok &= case("512 x 12.5M shard", q3, E3, None, None, 50, prep=True)
        del E3
        q4 = torch.randn(512, 128, device="cuda", generator=g); E4 = torch.randn(6_000_000, 128, device="cuda", generator=g)
        ok &= case("512 x 6M, D=128", q4, E4, None, None, 50, prep=True)
        del E4
        q5 = torch.randn(4096, 128, device="cuda", generator=g); E5 = torch.randn(50_000, 128, device="cuda", generator=g)
        sp5, si5 = seen(4096, 50_000, 16, g)
        ok &= case("4096 x 50k, D=128", q5, E5, sp5, si5, 50)
    print("ALL IDENTICAL" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
