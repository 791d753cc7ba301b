#!/usr/bin/env python3
"""Secondary legs of bench.py, each run in a CHILD process (`python bench_legs.py --leg NAME`) that prints one JSON object: whatever
goes wrong in a leg (memory, a time-out) must not take the headline line with it.

    config1   MF-BPR d=64 on Amazon2014Beauty shapes, B = 2048 triplets            (BASELINE.json configs[0]; MF-BPR/main.py:81-131)
    config3   LightGCN d=64, 3 layers, on a Yelp2018-shaped graph, B = 2048         (configs[2]; LightGCN/main.py:77-172) + SpMM roofline
    config4   DeepFM on the synthetic Amazon2023Games context schema, B = 4096      (configs[3]; DeepFM/main.py:201-276) + field-bag roofline
    config5   SASRec d=128 on the synthetic 100 M-item table, one GPU               (configs[4]; 64 DISTINCT batches)

Each GPU number has the CPU oracle of the same step timed beside it on this box's host cores (`cpu_baseline`: kind "port", the thread
count probed and reported) on a bounded sample.  Synthetic inputs as SURVEY.md section 8d defines them (seed 1)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
MFMA_F32_PEAK_TF = 157.3      # MI355X fp32 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
IC_PEAK_GBS = 8600.0          # Infinity Cache, random 1-KB-class rows of a resident table (MI355X_MICROARCH.md "Indexed rows: gather into LDS")


def ev_ms(fn, iters=30, warmup=5):
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def wall_ms(fn, iters=50, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def cpu_steps(step, unit_per_step, unit, what, budget_s=8.0, max_steps=200):
    """Time `step()` (the oracle's forward + backward + Adam) on the host cores: the fastest of a few thread counts, then a bounded run."""
    ncpu = os.cpu_count() or 1
    best, cores = None, 1
    for t in sorted({c for c in (4, 8, 16, 32) if c <= ncpu} | {min(ncpu, 8)}):
        torch.set_num_threads(t)
        step()
        t0 = time.time()
        step()
        d = time.time() - t0
        if best is None or d < best:
            best, cores = d, t
        if d > 2.0:
            break
    torch.set_num_threads(cores)
    t0, n = time.time(), 0
    while time.time() - t0 < budget_s and n < max_steps:
        step()
        n += 1
    dt = time.time() - t0
    return {"value": round(n * unit_per_step / dt, 1), "unit": unit, "cores": cores, "kind": "port",
            "sample": f"{n} steps of {what}, {dt:.1f} s, torch {torch.__version__} CPU, {cores} threads"}


# ------------------------------------------------------------------------------------------------ config 1: MF-BPR / Beauty
def leg_config1():
    from oracle import mf as omf
    from recboard_amd.gen import MFEngine
    U, N, B, D = 22363, 12101, 2048, 64
    rng = np.random.default_rng(1)
    w = 1.0 / np.arange(1, N + 1)
    w /= w.sum()
    bs = [(rng.integers(0, U, (B, 1)), rng.choice(N, (B, 1), p=w), rng.integers(0, N, (B, 1))) for _ in range(16)]
    dev = [tuple(torch.from_numpy(a).cuda() for a in b) for b in bs]
    m = MFEngine(U, N, D, lr=1e-3, weight_decay=1e-6)
    it = iter(range(10 ** 9))
    ms_eager = wall_ms(lambda: m.train_step(*dev[next(it) % 16]), iters=100, warmup=20)
    ms = wall_ms(lambda: m.train_step_graph(*dev[next(it) % 16]), iters=300, warmup=20)
    # CPU oracle: dense tables, autograd, dense Adam with coupled L2 (what the reference's torch path executes)
    g = torch.Generator().manual_seed(1)
    Ut = (torch.randn(U, D, generator=g) * 1e-4).requires_grad_(True)
    It = (torch.randn(N, D, generator=g) * 1e-4).requires_grad_(True)
    opt = torch.optim.Adam([Ut, It], lr=1e-3, weight_decay=1e-6)
    cb = [tuple(torch.from_numpy(a) for a in b) for b in bs]
    k = iter(range(10 ** 9))

    def cpu_step():
        u, p, n = cb[next(k) % 16]
        opt.zero_grad()
        omf.fit(Ut, It, u, p, n).backward()
        opt.step()
    ms_graph, ms = ms, min(ms, ms_eager)        # (`value` is the faster form: whichever the Coach would run)
    # the step's dominant launch: scatter-add of the 3 x B gradient rows + the DENSE Adam of all U + N rows (coupled L2 touches every row:
    # MF-BPR/main.py:115-131 with torch.optim.Adam(weight_decay)) = re_scatter_adam_rows_small.  Algorithmic bytes: per parameter element
    # 12 B read (p, m, v) + 12 B written, per contribution row 4 (key) + 4 D read.
    t_own = None
    try:
        from recboard_amd import ops
        u0, p0, n0 = dev[0]
        m.train_step(u0, p0, n0)
        own = getattr(m, "_owner_call", None)
        if own is not None:
            t_own = ev_ms(own, iters=100)
    except Exception:  # noqa: BLE001
        t_own = None
    own_bytes = (U + N) * D * 24 + 3 * B * (4 + 4 * D)
    roof1 = None
    if t_own:
        roof1 = {"kernel": "scatter_adam_owner_k (re_scatter_adam_rows_small): scatter-add of the step's 3 x B gradient rows + dense Adam of all "
                           f"{U + N} rows in one launch", "bound": "infinity_cache", "achieved": round(own_bytes / (t_own * 1e-3) / 1e9, 1), "peak": IC_PEAK_GBS,
                 "unit": "GB/s", "frac": round(own_bytes / (t_own * 1e-3) / 1e9 / IC_PEAK_GBS, 4), "traffic": None, "launch_ms": round(t_own, 4),
                 "frac_of_hbm_peak": round(own_bytes / (t_own * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "peak_source": "MI355X_MICROARCH.md 'Indexed rows': 8.6 TB/s chip-wide from a 38 MB Infinity-Cache-resident table (the guide's one measured Infinity Cache rate)",
                 "work": f"algorithmic {own_bytes / 1e6:.1f} MB per launch: {(U + N) * D} parameter elements x (12 B read + 12 B written) + {3 * B} "
                         f"contribution rows x (4 + {4 * D}) B (the 26 MB of parameters and moments live in the Infinity Cache between steps: "
                         "the bound a launch of this size sees is the cache's, not HBM's)"}
    return {"metric": "train triplets/sec (MF-BPR d=64, Beauty shapes, B=2048, 1 GPU)", **({"roofline": roof1} if roof1 else {}), "value": round(B / (ms * 1e-3), 1), "unit": "triplets/s",
            "ms_per_step": round(ms, 4), "ms_per_step_graph": round(ms_graph, 4), "config": {"workload": f"MF-BPR d=64, {U} users x {N} items, B={B}, Adam(lr 1e-3, wd 1e-6); users uniform, positives Zipf(1.0)"},
            "ms_per_step_eager": round(ms_eager, 4),
            "launch": "per step: the step's scalars (one tiny launch), the fused triplet forward + backward (gradient rows + their destination rows in the "
                      "user | item arena), ONE owner-computes launch = scatter-add + dense Adam of all 34 464 rows; eager launches or one hipGraph replay, "
                      "whichever is faster (`value`)",
            "cpu_baseline": cpu_steps(cpu_step, B, "triplets/s", f"B={B} (oracle/mf.py fit + backward + torch.optim.Adam)")}


# ------------------------------------------------------------------------------------------------ config 3: LightGCN / Yelp2018 shape
def yelp_graph(rng):
    U, N, E = 77277, 45638, 1949342
    deg_u = np.clip(rng.lognormal(2.6, 1.0, U), 1, 2000)
    deg_u = deg_u / deg_u.sum()
    wi = 1.0 / np.arange(1, N + 1) ** 0.8
    wi /= wi.sum()
    eu, ei = rng.choice(U, int(E * 1.08), p=deg_u), rng.choice(N, int(E * 1.08), p=wi)
    key = np.unique(eu.astype(np.int64) * N + ei)[:E]
    return U, N, key // N, key % N, wi


def leg_config3():
    from oracle import lightgcn as olg
    from recboard_amd.gen import LightGCNEngine
    from recboard_amd.graph import to_normalized_adj
    rng = np.random.default_rng(1)
    U, N, eu, ei, wi = yelp_graph(rng)
    B, D, Ly = 2048, 64, 3
    crow, col, val = to_normalized_adj(U, N, eu, ei)
    nnz = len(col)
    lg = LightGCNEngine(U, N, crow, col, val, D, Ly, lr=1e-3, weight_decay=1e-3)
    with torch.no_grad():
        for q in lg.params.values():
            q.normal_(0, 0.1)
    bs = [(rng.integers(0, U, (B, 1)), rng.choice(N, (B, 1), p=wi), rng.integers(0, N, (B, 1))) for _ in range(8)]
    dev = [tuple(torch.from_numpy(a).cuda() for a in b) for b in bs]
    it = iter(range(10 ** 9))
    ms_eager = wall_ms(lambda: lg.train_step(*dev[next(it) % 8]), iters=20, warmup=5)
    ms = wall_ms(lambda: lg.train_step_graph(*dev[next(it) % 8]), iters=50, warmup=5)
    t_sp = ev_ms(lambda: lg._spmm(lg.X0, lg.Xa))
    rows = U + N
    # per non-zero 4 (val) + 8 (col) B streamed from HBM + one 4 D-byte X row (cache-resident: 31.5 MB < 256 MB Infinity
    # Cache, so its HBM share is one read of X); per output row 8 (crow) + 4 D written
    alg = nnz * 12 + rows * (8 + 4 * D) + rows * 4 * D
    gbs = alg / (t_sp * 1e-3) / 1e9
    # CPU oracle on a bounded sample: the same graph, autograd through 3 + 3 SpMMs, dense Adam without weight decay
    g = torch.Generator().manual_seed(1)
    Ut = (torch.randn(U, D, generator=g) * 0.1).requires_grad_(True)
    It = (torch.randn(N, D, generator=g) * 0.1).requires_grad_(True)
    opt = torch.optim.Adam([Ut, It], lr=1e-3)
    c_crow, c_col, c_val = (np.asarray(a.cpu()) if torch.is_tensor(a) else np.asarray(a) for a in (crow, col, val))
    k = iter(range(10 ** 9))

    def cpu_step():
        u, p, n = bs[next(k) % 8]
        opt.zero_grad()
        rec, emb = olg.fit(Ut, It, c_crow, c_col, c_val, torch.from_numpy(u).reshape(-1), torch.from_numpy(p).reshape(-1), torch.from_numpy(n).reshape(-1), Ly)
        (rec + 1e-3 * emb).backward()
        opt.step()
    return {"metric": "train triplets/sec (LightGCN d=64 L=3, Yelp2018 shapes, B=2048, 1 GPU)", "value": round(B / (min(ms, ms_eager) * 1e-3), 1), "unit": "triplets/s",
            "ms_per_step": round(min(ms, ms_eager), 4), "ms_per_step_graph": round(ms, 4), "ms_per_step_eager": round(ms_eager, 4),
            "launch": "one hipGraph replay per step, or its launches issued eagerly: `value` is the faster form",
            "config": {"workload": f"LightGCN d=64, 3 layers, {U} users x {N} items, {len(eu)} edges (adjacency nnz {nnz}), B={B}: 3 + 3 SpMMs per step, "
                                   "loss = rec + 1e-3 emb, Adam without weight decay (LightGCN/main.py:139-160)"},
            # the bound that binds: every non-zero gathers a 256-B row of the 31.5 MB X out of the Infinity Cache (round-5 verdict: a launch whose
            # operand lives in a cache is priced against that cache's measured ceiling, not against HBM on bytes that never leave the die)
            "roofline": {"kernel": "spmm_csr_rows / spmm_csr_long (re_spmm_csr): one propagation X <- A X", "bound": "infinity_cache",
                         "achieved": round((alg + nnz * 4 * D) / (t_sp * 1e-3) / 1e9, 1), "peak": IC_PEAK_GBS, "unit": "GB/s",
                         "frac": round((alg + nnz * 4 * D) / (t_sp * 1e-3) / 1e9 / IC_PEAK_GBS, 4), "traffic": None, "launch_ms": round(t_sp, 4),
                         "algorithmic_hbm": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                             "note": "algorithmic bytes only (CSR stream + each X / Y row once) against the HBM peak"},
                         "gathered_rows": {"achieved": round((alg + nnz * 4 * D) / (t_sp * 1e-3) / 1e9, 1), "peak": 8600.0, "unit": "GB/s",
                                           "frac": round((alg + nnz * 4 * D) / (t_sp * 1e-3) / 1e9 / 8600.0, 4),
                                           "note": "what bounds the launch: every non-zero gathers a 256-B X row from the 31.5 MB X, which lives in the "
                                                   "Infinity Cache -- MI355X_MICROARCH.md measures 8.6 TB/s for uniformly random rows of a 38 MB table"},
                         "work": f"algorithmic {alg / 1e6:.1f} MB per launch: {nnz} non-zeros x 12 B + {rows} rows x (8 + {4 * D} written + {4 * D} read once); "
                                 f"with every gathered X row counted ({nnz} x {4 * D} B, served by L2 / Infinity Cache): {(alg + nnz * 4 * D) / (t_sp * 1e-3) / 1e9:.0f} GB/s; "
                                 f"{2 * D * nnz / (t_sp * 1e-3) / 1e9:.0f} GFLOP/s"},
            "cpu_baseline": cpu_steps(cpu_step, B, "triplets/s", f"B={B} on the same graph (oracle/lightgcn.py fit + backward + Adam)", budget_s=12.0, max_steps=40)}


# ------------------------------------------------------------------------------------------------ config 4: DeepFM / Games context schema
def leg_config4():
    from oracle import deepfm as odf
    from recboard_amd import ops
    from recboard_amd.deepfm import DeepFMEngine
    # no DeepFM config exists for Amazon2023Games in the reference -- synthetic schema: USER, ITEM (benchmark
    # cardinalities) + 8 context fields
    counts = [94762, 25612, 7, 24, 12, 5, 50, 500, 5000, 50000]
    B, D, hid = 4096, 10, (400, 400, 400)
    rng = np.random.default_rng(1)
    d = DeepFMEngine(counts, D, hid, batch_norm=True, hidden_dropout_rate=0.1, lr=1e-3, embedding_decay=0.05)
    bs = [(np.stack([rng.integers(0, c, B) for c in counts], 1), (rng.random((B, 1)) < 0.3).astype(np.int64)) for _ in range(8)]
    dev = [(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()) for x, y in bs]
    it = iter(range(10 ** 9))
    ms_eager = wall_ms(lambda: d.train_step(*dev[next(it) % 8]), iters=30, warmup=10)
    ms = wall_ms(lambda: d.train_step_graph(*dev[next(it) % 8]), iters=100, warmup=10)
    x0 = dev[0][0]
    t_bag = ev_ms(lambda: ops.fm_bag_fwd(d.T, d.TL.reshape(-1), d.bias, d.offsets, x0), iters=100)
    # the step's dominant kernels: the MLP's products (DeepFM/main.py:103-124,151-164) -- the 400 x 400 layer's forward / input-gradient /
    # weight-gradient forms, timed alone (re_gemm_f32: exact fp32 on v_mfma_f32_16x16x4_f32)
    gh = torch.randn(B, 400, device="cuda")
    gw = torch.randn(400, 400, device="cuda")
    gz = torch.randn(B, 400, device="cuda")
    gemm_ms = {"forward x W^T [4096 x 400 x 400]": ev_ms(lambda: ops.gemm(gh, gw, transB=True), iters=50),
               "input gradient dz W [4096 x 400 x 400]": ev_ms(lambda: ops.gemm(gz, gw), iters=50),
               "weight gradient dz^T x [400 x 400 x 4096]": ev_ms(lambda: ops.gemm(gz, gh, transA=True), iters=50)}
    gfl = 2.0 * B * 400 * 400
    t_g = gemm_ms["forward x W^T [4096 x 400 x 400]"]
    step_fl = 3 * 2.0 * B * (100 * 400 + 400 * 400 + 400 * 400)                     # forward + input gradient + weight gradient of the three layers
    F = len(counts)
    alg = B * F * (8 + 4 * D + 4) + B * (F * D * 4 + 4)                      # F (8 + 40 + 4) B gathered per row, + E [B, F D] and fm_lr [B] written
    gbs = alg / (t_bag * 1e-3) / 1e9
    # CPU oracle: per-field tables, autograd, BatchNorm in training mode, two Adam groups as the reference builds them
    g = torch.Generator().manual_seed(1)
    tables = [(torch.randn(c, D, generator=g) * 1e-4).requires_grad_(True) for c in counts]
    tables_lr = [(torch.randn(c, 1, generator=g) * 1e-4).requires_grad_(True) for c in counts]
    lr_bias = torch.zeros(1, requires_grad=True)
    dims = [F * D] + list(hid)
    mlp = []
    for i in range(len(hid)):
        mlp.append({"linear.weight": (torch.randn(dims[i + 1], dims[i], generator=g) * (2.0 / (dims[i] + dims[i + 1])) ** 0.5).requires_grad_(True),
                    "linear.bias": torch.zeros(dims[i + 1], requires_grad=True), "bn.weight": torch.ones(dims[i + 1], requires_grad=True),
                    "bn.bias": torch.zeros(dims[i + 1], requires_grad=True)})
    mlp.append({"weight": (torch.randn(1, dims[-1], generator=g) * 0.05).requires_grad_(True), "bias": torch.zeros(1, requires_grad=True)})
    emb = tables + tables_lr
    other = [lr_bias] + [v for blk in mlp for v in blk.values()]
    opt = torch.optim.Adam([{"params": emb, "weight_decay": 0.05}, {"params": other, "weight_decay": 0.0}], lr=1e-3)
    k = iter(range(10 ** 9))

    def cpu_step():
        x, y = bs[next(k) % 8]
        opt.zero_grad()
        odf.fit(tables, tables_lr, lr_bias, mlp, torch.from_numpy(x), torch.from_numpy(y).float()).backward()
        torch.nn.utils.clip_grad_norm_(emb + other, 10.0)
        opt.step()
    return {"metric": "train rows/sec (DeepFM, synthetic Amazon2023Games context schema, B=4096, 1 GPU)", "value": round(B / (min(ms, ms_eager) * 1e-3), 1), "unit": "rows/s",
            "ms_per_step": round(min(ms, ms_eager), 4), "ms_per_step_graph": round(ms, 4), "ms_per_step_eager": round(ms_eager, 4),
            "launch": "one hipGraph replay per step, or its launches issued eagerly: `value` is the faster form",
            "config": {"workload": f"DeepFM: {F} embedding fields (cardinalities {counts}), D={D}, MLP {dims}->1 with BatchNorm + dropout 0.1, B={B}, "
                                   "BCE, clip 10, Adam with the reference's two weight-decay groups (DeepFM/main.py:187-199,264-268)"},
            "roofline": {"kernel": "gemm_wide_k (re_gemm_f32): the MLP's 400 x 400 layer, forward product x W^T at [4096 x 400 x 400] -- the step's dominant "
                                   "kernel family (9 such products a step: 3 layers x forward / input gradient / weight gradient)", "bound": "mfma",
                         "achieved": round(gfl / (t_g * 1e-3) / 1e12, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                         "frac": round(gfl / (t_g * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4), "traffic": None, "launch_ms": round(t_g, 4),
                         "forms": {k: {"launch_ms": round(v, 4), "TFLOP/s": round(gfl / (v * 1e-3) / 1e12, 2), "frac": round(gfl / (v * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4)}
                                   for k, v in gemm_ms.items()},
                         "whole_step": {"mlp_FLOP": step_fl, "TFLOP/s": round(step_fl / (min(ms, ms_eager) * 1e-3) / 1e12, 2),
                                        "frac": round(step_fl / (min(ms, ms_eager) * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
                                        "note": "the MLP's FLOP over the WHOLE step time (embedding bag, BatchNorm passes, scatter-add, clip, Adam included)"},
                         "work": f"2 M N K = {gfl:.3e} FLOP per launch, exact fp32 (v_mfma_f32_16x16x4_f32); 64-row x 112-column workgroup tiles: 256 workgroups, "
                                 "7 column tiles a wave against 6.25 ideal"},
            "roofline_fm_bag": {"kernel": "fm_bag_fwd_k (re_fm_bag_fwd): every field's row + FM second-order term + LR term per input row", "bound": "hbm",
                         "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                         "launch_ms": round(t_bag, 4),
                         "work": f"algorithmic {alg / 1e6:.2f} MB per launch: {B} rows x {F} fields x (8 + {4 * D} + 4) B gathered + {F * D * 4 + 4} B written per row "
                                 "(at 2.3 MB a launch is latency-bound: the tables' hot rows are L2-resident)"},
            "cpu_baseline": cpu_steps(cpu_step, B, "rows/s", f"B={B} (oracle/deepfm.py fit + backward + clip + Adam)")}


# ------------------------------------------------------------------------------------------------ config 5: SASRec d=128 / 100 M items, one GPU
def c5_batches(rng, nb, N, B, S):
    out = []
    for _ in range(nb):
        lens = np.clip(rng.geometric(1 / 5.9, B) + 1, 1, S - 1)
        seq = np.zeros((B, S), np.int64)
        for b in range(B):
            seq[b, S - lens[b]:] = np.minimum(rng.zipf(1.05, lens[b]), N)
        pos = np.where(seq > 0, np.minimum(rng.zipf(1.05, (B, S)), N) - 1, 0)
        neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
        out.append(tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg)))
    return out


def leg_config5(steps=128, warmup=16, nbatch=64):
    """BASELINE.json configs[4] on this GPU: SASRec d = 128, L = 2, maxlen 50, BCE, on the synthetic 100 000 000-item table (SURVEY.md
    section 8d C5: item popularity Zipf(1.05), B = 512, one uniform negative, table ~ N(0, 0.02^2) from the counter-based generator).  The table and
    its two Adam moment tables (154 GB) live in HBM; the step is one batch-preparation launch + one hipGraph replay.  64 DISTINCT batches
    are cycled: their uniform negatives alone touch 64 x ~3.5 k x 3 x 512 B ~ 350 MB of table rows, beyond the 256 MB Infinity Cache."""
    N, D, B, S = int(os.environ.get("RECBENCH_C5_ITEMS", 100_000_000)), 128, 512, 50      # (the override: tests/test_gpu_bench.py runs the leg small)
    try:
        torch.cuda.empty_cache()
        free, _ = torch.cuda.mem_get_info()
        need = 3 * (N + 1) * D * 4 + (8 << 30)
        if free < need:
            return {"skipped": f"needs {need / 1e9:.0f} GB of HBM, {free / 1e9:.0f} GB free"}
        from recboard_amd.large import SASRecLargeTableEngine
        t0 = time.time()
        eng = SASRecLargeTableEngine(N, S, D, 2, dropout_rate=0.5, loss="BCE", lr=1e-3, weight_decay=1e-6, seed=1)
        torch.cuda.synchronize()
        t_init = time.time() - t0
        bs = c5_batches(np.random.default_rng(1), nbatch, N, B, S)
        # (an epoch loop has the next batch at hand: it is prepared by jobs of this step's tail launch -- SASRecLargeTableEngine._train_step_graph_tail)
        for i in range(warmup):
            eng.train_step_graph(*bs[i % nbatch], next_batch=bs[(i + 1) % nbatch])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = eng.train_step_graph(*bs[(warmup + i) % nbatch], next_batch=bs[(warmup + i + 1) % nbatch])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        eng.check_handover()
        free2, total = torch.cuda.mem_get_info()
        traffic = None
        try:
            import glob
            latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), key=lambda q: int(os.path.basename(q)[1:].split("_")[0]))[-1]
            with open(latest) as f:
                traffic = json.load(f).get("config5_step")
            if traffic is not None:
                traffic["from"] = "profiles/" + os.path.basename(latest)
        except Exception:  # noqa: BLE001
            pass
        out = {"metric": f"train samples/sec (SASRec d=128 on the synthetic {N / 1e6:g} M-item table, B=512, 1 GPU)", "value": round(B / dt, 1),
               "unit": "samples/s", "ms_per_step": round(dt * 1e3, 4), "steps": steps, "warmup": warmup, "distinct_batches": nbatch,
               "final_loss": round(float(loss), 5), "table": f"{N + 1} x {D} fp32 + two Adam moment tables",
               "hbm_used_GB": round((total - free2) / 1e9, 1), "table_init_s": round(t_init, 1), "hbm_traffic_per_step": traffic,
               "launch": "one stage launch + one hipGraph replay per step; the next batch is prepared by jobs of the step's tail launch",
               "data": "synthetic: Zipf(1.05) item popularity, lengths ~ clip(Geometric(mean 5.9) + 1, 1, 49)"}
        del eng, bs
        torch.cuda.empty_cache()
        return out
    except Exception as e:  # noqa: BLE001  (the headline line must not depend on this leg)
        torch.cuda.empty_cache()
        return {"skipped": f"{type(e).__name__}: {str(e)[:200]}"}


# ------------------------------------------------------------------------------------------------ config 5, row-sharded (all ranks call this)
def config5_sharded(dist, rank, world, local, steps=40, warmup=8, nbatch=32):
    """BASELINE.json configs[4] as named: the 100 M x 128 table ROW-SHARDED over the ranks (rows r mod G), B = 512 per GPU (weak scaling):
    one all-to-all round trip for the batch's rows, one all-to-all of gradient rows to their owners + row-sparse Adam there, one all-reduce
    of the encoder's gradient arena; fixed-capacity exchanges (no host sync), the whole step one hipGraph replay with the RCCL collectives
    inside (recboard_amd.large.SASRecShardedEngine).  Every rank runs it; rank 0 gets the result."""
    N, D, B, S = int(os.environ.get("RECBENCH_C5_ITEMS", 100_000_000)), 128, 512, 50
    try:
        from recboard_amd.large import SASRecShardedEngine
        assert dist.get_world_size() == world
        t0 = time.time()
        model = SASRecShardedEngine(N, S, D, 2, dropout_rate=0.5, loss="BCE", lr=1e-3, weight_decay=1e-6, seed=1, device=f"cuda:{local}", capacity_factor=0.3)
        torch.cuda.synchronize()
        t_init = time.time() - t0
        bs = c5_batches(np.random.default_rng(1 + rank), nbatch, N, B, S)
        for i in range(warmup):
            model.train_step_graph(*bs[i % nbatch])
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = model.train_step_graph(*bs[(warmup + i) % nbatch])
        overflowed = model.settle_overflow()        # (steps whose exchange overflowed were no-ops and have been re-run on the exact-size path: inside the timed region)
        torch.cuda.synchronize()
        dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], device="cuda", dtype=torch.float64)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dt = float(dt.item()) / steps
        out = {"metric": "train samples/sec (SASRec d=128, 100 M-item table row-sharded over the GPUs, B=512/GPU)", "value": round(world * B / dt, 1),
               "unit": "samples/s", "n_gpus": world, "ms_per_step": round(dt * 1e3, 4), "steps": steps, "warmup": warmup, "distinct_batches_per_rank": nbatch,
               "scaling": "weak", "final_loss_rank0": round(float(loss), 5), "rows_per_rank": model.table.local_rows,
               "table_GB_per_rank": round(3 * model.table.local_rows * D * 4 / 1e9, 1), "table_init_s": round(t_init, 1),
               "launch": "one batch-preparation launch + one hipGraph replay per step (fixed-capacity exchanges, factor 0.3: no host sync)",
               "steps_rerun_on_the_exact_path": overflowed, "world_size": dist.get_world_size(), "backend": dist.get_backend()}
        model.release_graphs()
        del model, bs
        torch.cuda.empty_cache()
        return out
    except Exception as e:  # noqa: BLE001
        return {"skipped": f"{type(e).__name__}: {str(e)[:300]}"}


LEGS = {"config1": leg_config1, "config3": leg_config3, "config4": leg_config4, "config5": leg_config5}


def run_child(leg, timeout=420):
    """-> the child's JSON object, or {"skipped": reason}."""
    import subprocess
    try:
        torch.cuda.empty_cache()
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", leg], capture_output=True, text=True, timeout=timeout)
        for ln in reversed(r.stdout.splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"skipped": f"child exited with {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
    except Exception as e:  # noqa: BLE001
        return {"skipped": f"{type(e).__name__}: {str(e)[:200]}"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--leg", required=True, choices=sorted(LEGS))
    a = ap.parse_args()
    torch.cuda.set_device(0)
    print(json.dumps(LEGS[a.leg]()), flush=True)
