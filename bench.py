#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): SASRec d=64, L=2, 1 head, maxlen 50, BCE, dropout 0.5, Adam(lr 5e-4, wd 1e-6),
B=512 sequences per GPU per step, on synthetic data shaped like Amazon2014Beauty_550_LOU (22 363 users, 12 101 items;
SURVEY.md §8d C2).  A "step" = one full training step from a RAW (seq, pos, neg) batch already resident in HBM: the batch
preparation launch (mask of the non-pad positions, their number, scatter destination rows, the encoder's work plan -- what the
reference does at the top of `fit`, SASRec/main.py:199-204 -- plus staging into the captured step's buffers), then forward,
backward, dense Adam.  `value` = training sequences per second over all GPUs (weak scaling: 512 per GPU).  `coach_loop` is the
same step driven by the engine's Coach from HOST batches (H2D copies and the epoch loop's Python included).
The second half of BASELINE's metric -- full-catalog items scored per second -- is measured in the same run, outside
the timed region, over all 22 363 users x 12 101 items with the fused score+mask+top-K kernel, and reported in
`items_scored_per_sec` and `roofline_score` (MFMA-bound).  `roofline` is the dominant launch group of the timed region (the
encoder step: forward + criterion + backward of every work item in one kernel, then the weight gradients; MFMA-bound fp32),
`roofline_gather` the HBM-bound embedding gather.  `cpu_baseline` times the torch-CPU oracle of the same training step on this box's host cores;
`eval_baselines` times the evaluation as the reference executes it (dense scores, masked fill, torch.topk) through ROCm aten
on the same GPU and through torch on the host cores; `train_baseline_aten_gpu` a torch.nn SASRec step (eager ROCm aten) on the
same GPU.  `config5` (N = 1 runs only) is BASELINE.json's configs[4] on this GPU: the same step at d = 128 on the synthetic
100 000 000-item table with the row-sparse Adam (154 GB of HBM, 64 distinct batches; `--no-c5` skips it); at N > 1 `config5_sharded` is
that table row-sharded over the ranks.  `config1` / `config3` / `config4` (N = 1) are the other BASELINE configs' steps with their own
CPU-oracle baselines and the SpMM / field-bag rooflines (bench_legs.py, child processes); `sampler` the device sampler's rate alone
and feeding the Coach.  `--gpus N` without a launcher starts the N ranks itself (torch.distributed.run on 127.0.0.1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BEAUTY = dict(users=22363, items=12101, D=64, L=2, S=50, B=512, p_drop=0.5, lr=5e-4, wd=1e-6)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32 matrix peak


def synth_batches(cfg, nbatch, seed):
    """lengths ~ clip(Geometric(mean 5.9)+1, 1, 49), items Zipf(1.0), left-padded, ids +1."""
    rng = np.random.default_rng(seed)
    N, B, S = cfg["items"], cfg["B"], cfg["S"]
    w = 1.0 / np.arange(1, N + 1)
    w /= w.sum()
    out = []
    for _ in range(nbatch):
        lens = np.clip(rng.geometric(1.0 / 5.9, B) + 1, 1, S - 1)
        seq = np.zeros((B, S), np.int64)
        pos = np.zeros((B, S), np.int64)
        neg = np.zeros((B, S), np.int64)
        for b in range(B):
            L = lens[b]
            seq[b, S - L:] = rng.choice(N, L, p=w) + 1
            pos[b, S - L:] = rng.choice(N, L, p=w)
            neg[b, S - L:] = rng.integers(0, N, L)
        out.append((seq, pos, neg))
    return out


def large_batch_roofline(cfg, B=8192, steps=40):
    """The D = 64 step in its THROUGHPUT regime: the same model at B = 8 192 Beauty-shaped sequences per step (~16 tiles of real tokens per
    CU instead of ~1; what a rank sees when a job scales its batch): samples/s and the executed-FLOP fraction of the fp32 matrix peak.
    The plan hands a batch to whichever of the two encoder kernels is faster for it (csrc/enc_plan_body.h; scripts/large_batch.py,
    scripts/long_mix.py): the one-tile-per-workgroup kernel (one workgroup per CU, further tiles from a counter) up to ~10 tiles per CU --
    B = 2 048 here -- and the fp32 workgroup-per-item kernel beyond -- B = 8 192."""
    from recboard_amd.sasrec import SASRecEngine
    big = dict(cfg, B=B)
    m = SASRecEngine(cfg["items"], cfg["S"], cfg["D"], cfg["L"], dropout_rate=cfg["p_drop"], loss="BCE", lr=cfg["lr"], weight_decay=cfg["wd"], seed=1)
    bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in synth_batches(big, 4, seed=11)]
    # (launched as the headline is, and as an epoch loop does: the NEXT batch is handed to the step, whose tail launch prepares it -- in front of
    #  the step the preparation of 4 096 sequences is an 80 us launch of its own: 0.486 -> 0.421 ms)
    for i in range(8):
        m.train_step_graph(*bs[i % 4], next_batch=bs[(i + 1) % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        m.train_step_graph(*bs[i % 4], next_batch=bs[(i + 1) % 4])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    m.check_handover()
    hdr = m.prepare_batch(*bs[0]).plan.view(torch.int32)[:8].cpu().numpy()
    n_items, n_tiles, D, L = int(hdr[0]), int(hdr[1]), cfg["D"], cfg["L"]
    fl_exec = L * n_tiles * (24 * 2 * 16 * D * D + 6 * 2 * 16 * 16 * D)           # (the same count as `roofline.work`: per tile and block)
    tf = fl_exec / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4),
            "samples_per_sec": round(B / (ms * 1e-3), 1), "ms_per_step": round(ms, 4), "B": B, "tiles": n_tiles, "work_items": n_items,
            "tiles_per_cu": round(n_tiles / 256.0, 1),
            "kernel": "the one-tile-per-workgroup step (enc_tile_step_k)" if int(hdr[7]) == 1 else "the fp32 workgroup-per-item step (enc_step_k<64>: the plan's "
                      "choice beyond ~10 tiles or ~1.5 chained tiles per CU)",
            "work": f"executed FLOP as in `roofline.work`: {fl_exec:.3e} per step on {n_tiles} tiles; whole step (preparation, encoder, tail, Adam) per replay",
            "launch": "train_step_graph(next_batch=...): one stage launch + one replay per step, the next batch prepared by jobs of this step's tail launch"}


def fp32_exact_step(cfg, steps=200):
    """The headline configuration with every linear map as EXACT fp32 products: the same engine with `tile_step = False`, i.e. the
    workgroup-per-item kernels (enc_step_k<64>: v_mfma_f32_16x16x4_f32, bitwise an fmaf chain) instead of the tile kernel's three-bf16-product
    form.  The like-precision companion of `value` (the reference computes in fp32)."""
    from recboard_amd.sasrec import SASRecEngine
    m = SASRecEngine(cfg["items"], cfg["S"], cfg["D"], cfg["L"], dropout_rate=cfg["p_drop"], loss="BCE", lr=cfg["lr"], weight_decay=cfg["wd"], seed=1)
    m.tile_step = False
    bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in synth_batches(cfg, 8, seed=1)]
    for i in range(20):
        m.train_step_graph(*bs[i % 8], next_batch=bs[(i + 1) % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        m.train_step_graph(*bs[i % 8], next_batch=bs[(i + 1) % 8])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    m.check_handover()
    return {"ms_per_step": round(ms, 4), "samples_per_sec": round(cfg["B"] / (ms * 1e-3), 1), "steps": steps,
            "kernel": "enc_step_k<64> (forward + criterion + backward of a work item, exact fp32 MFMA) + the same tail launches",
            "dtype": "f32 throughout (storage, products, accumulation)"}


def event_time_ms(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def cpu_baseline(cfg, batches, budget_s=15.0):
    """The oracle's torch-CPU restatement of the same step (fit + backward + dense Adam), all host cores."""
    from oracle import sasrec as osas
    from recboard_amd.sasrec import param_shapes
    ncpu = os.cpu_count() or 1
    g = torch.Generator().manual_seed(1)
    P = {k: (torch.randn(s, generator=g) * 0.05).requires_grad_(True)
         for k, s in param_shapes(cfg["items"], cfg["S"], cfg["D"], cfg["L"]).items()}
    for k in P:
        if "LN" in k and k.endswith("weight"):
            P[k].data.fill_(1.0)
    opt = torch.optim.Adam(list(P.values()), lr=cfg["lr"], weight_decay=cfg["wd"])
    tb = [tuple(torch.from_numpy(a) for a in b) for b in batches]

    def step(i):
        seq, pos, neg = tb[i % len(tb)]
        opt.zero_grad()
        loss = osas.fit(P, seq, pos, neg, "BCE", cfg["L"])
        loss.backward()
        opt.step()

    # B*S = 25 600 tokens x D = 64 is too small for hundreds of threads (one step took 44 s with 256 threads on the
    # GPU box); probe a few thread counts and keep the fastest -- `cores` reports the count actually used.
    best, cores = None, 1
    for t in sorted({c for c in (4, 8, 16, 32, 64) if c <= ncpu} | {min(ncpu, 8)}):
        torch.set_num_threads(t)
        step(0)
        t0 = time.time()
        step(1)
        d = time.time() - t0
        if best is None or d < best:
            best, cores = d, t
        if d > 3.0:
            break
    torch.set_num_threads(cores)
    t0 = time.time()
    n = 0
    while time.time() - t0 < budget_s and n < 200:
        step(n)
        n += 1
    dt = time.time() - t0
    return {"value": round(n * cfg["B"] / dt, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "dropout": "off on the CPU side (the GPU step runs dropout 0.5: extra work there, none here)",
            "sample": f"{n} training steps of B={cfg['B']} (same shapes), {dt:.1f} s, "
                      f"torch {torch.__version__} CPU, {cores} threads"}


def pmc_traffic(*kernels, algorithmic_bytes=None):
    """HBM bytes per launch of the named kernels (summed) from the committed PMC summary -- FETCH_SIZE / WRITE_SIZE cannot be
    read from inside the process, they come from separate rocprofv3 --pmc passes of this same command (profiles/).
    algorithmic_bytes: what the launch group must move at least; `wasted_traffic` = counted / algorithmic (1.0 = nothing re-read)."""
    import glob
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    found = sorted(glob.glob(os.path.join(root, "r*_pmc_traffic.json")), key=lambda p_: int(os.path.basename(p_)[1:].split("_")[0]))
    if not found:
        return None
    path = found[-1]                                   # the latest round's summary
    try:
        with open(path) as f:
            k = json.load(f)["kernels"]
        # a name is matched exactly, or as the prefix of an instantiation ("gather_rows_vec4<16" -> "gather_rows_vec4<16, 4, true, true>")
        hit = []
        for n in kernels:
            hit += [n] if n in k else [m for m in k if m.startswith(n) or ("::" + n) in m]      # (namespaced: "tl4::enc_tile_step_k")
        if not hit:
            return None
        tot = int(sum(k[n]["hbm_bytes_per_launch"] for n in hit))
        return {"hbm_bytes_per_launch": tot,
                **({"algorithmic_bytes": int(algorithmic_bytes), "wasted_traffic": round(tot / max(float(algorithmic_bytes), 1.0), 2)} if algorithmic_bytes else {}),
                "kernels": {n: k[n]["hbm_bytes_per_launch"] for n in hit},
                "source": f"profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of these kernels, separate passes; 2*FETCH+WRITE)"}
    except Exception:  # noqa: BLE001
        return None


def graph_time_ms(fn, reps=20, iters=10):
    """GPU time of `fn` (a few short launches): `reps` repetitions captured into one hipGraph, replayed `iters` times -- a loop of eager
    calls is bounded by the CPU launch path at these kernel sizes, not by the GPU."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    from recboard_amd.capture import recording
    g = torch.cuda.CUDAGraph()
    with recording(g, capture_error_mode="thread_local"):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (iters * reps)


def aten_train_baseline(cfg, batches, steps=30):
    """The training step as the reference runs it on a GPU: a torch.nn SASRec written the way SASRec/main.py:52-221 is (Embedding,
    nn.MultiheadAttention with a causal mask on LayerNorm'ed queries, Conv1d(k=1) feed-forward, BCE over the non-pad positions),
    fp32, torch.optim.Adam, eager ROCm aten kernels -- the baseline the engine's step replaces, on the same MI355X."""
    nn = torch.nn
    N, S, D, L, p = cfg["items"], cfg["S"], cfg["D"], cfg["L"], cfg["p_drop"]

    class FFN(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1, self.conv2 = nn.Conv1d(D, D, 1), nn.Conv1d(D, D, 1)
            self.d1, self.d2 = nn.Dropout(p), nn.Dropout(p)

        def forward(self, x):
            y = self.d2(self.conv2(torch.relu(self.d1(self.conv1(x.transpose(-1, -2)))))).transpose(-1, -2)
            return y + x

    class Model(nn.Module):
        def __init__(self):
            super().__init__()
            self.item, self.pos = nn.Embedding(N + 1, D, padding_idx=0), nn.Embedding(S, D)
            self.drop = nn.Dropout(p)
            self.aln = nn.ModuleList(nn.LayerNorm(D, eps=1e-8) for _ in range(L))
            self.att = nn.ModuleList(nn.MultiheadAttention(D, 1, dropout=p, batch_first=True) for _ in range(L))
            self.fln = nn.ModuleList(nn.LayerNorm(D, eps=1e-8) for _ in range(L))
            self.ffn = nn.ModuleList(FFN() for _ in range(L))
            self.last = nn.LayerNorm(D, eps=1e-8)
            self.register_buffer("mask", torch.ones(S, S, dtype=torch.bool).triu(1))

        def forward(self, seq, pos, neg):
            pad = (seq == 0).unsqueeze(-1)
            x = self.item(seq) * D ** 0.5 + self.pos(torch.arange(S, device=seq.device))
            x = self.drop(x).masked_fill(pad, 0.0)
            for l in range(L):
                qn = self.aln[l](x)
                x = self.att[l](qn, x, x, attn_mask=self.mask, need_weights=False)[0] + x
                x = self.ffn[l](self.fln[l](x)).masked_fill(pad, 0.0)
            u = self.last(x)
            keep = seq != 0
            u, E = u[keep], self.item.weight[1:]
            lp, ln = (u * E[pos[keep]]).sum(-1), (u * E[neg[keep]]).sum(-1)
            bce = torch.nn.functional.binary_cross_entropy_with_logits
            return bce(lp, torch.ones_like(lp)) + bce(ln, torch.zeros_like(ln))

    m = Model().cuda()
    opt = torch.optim.Adam(m.parameters(), lr=cfg["lr"], weight_decay=cfg["wd"])

    def step(i):
        seq, pos, neg = batches[i % len(batches)][:3]
        opt.zero_grad()
        m(seq, pos, neg).backward()
        opt.step()
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    return {"samples_per_sec": round(cfg["B"] / dt, 1), "ms_per_step": round(dt * 1e3, 3),
            "what": f"torch.nn SASRec (same shapes, fp32, eager ROCm aten, torch.optim.Adam), {steps} steps on the same GPU"}


def _latest_pmc_summary():
    import glob
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    found = sorted(glob.glob(os.path.join(root, "r*_pmc_traffic.json")), key=lambda p_: int(os.path.basename(p_)[1:].split("_")[0]))
    return found[-1] if found else None


def score_call_traffic():
    """HBM bytes of one whole re_score_topk call (all its launches) from the latest committed PMC summary (scripts/pmc_step.py)."""
    path = _latest_pmc_summary()
    try:
        with open(path) as f:
            c = json.load(f)["re_score_topk_call"]
        return {"hbm_bytes_per_launch": int(c["hbm_bytes_per_call"]), "algorithmic_lower_bound_bytes": int(c["algorithmic_lower_bound_bytes"]),
                "wasted_traffic": round(float(c["hbm_bytes_per_call"]) / max(float(c["algorithmic_lower_bound_bytes"]), 1.0), 2),
                "source": f"profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; 2*FETCH+WRITE, "
                          "summed over the launches of one call)"}
    except Exception:  # noqa: BLE001
        return None


def sampler_rates(cfg, model):
    """The device sampler alone (one re_seq_train_sample launch per batch: shuffled users, last-maxlen window, +1, left pad, one unseen uniform
    negative per position; SASRec/main.py:143-157) and the same sampler feeding the Coach's epoch loop (sample -> batch preparation -> graph
    replay, no host batch anywhere), on a synthetic Beauty-shaped training split."""
    from recboard_amd.coach import Coach
    from recboard_amd.sampler import DeviceInteractions, DeviceSeqSampler
    rng = np.random.default_rng(5)
    U, N, S, B = cfg["users"], cfg["items"], cfg["S"], cfg["B"]
    w = 1.0 / np.arange(1, N + 1)
    w /= w.sum()
    lens = np.clip(rng.geometric(1.0 / 5.9, U) + 2, 2, 200)
    ptr = np.zeros(U + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    inter = DeviceInteractions(ptr, rng.choice(N, int(ptr[-1]), p=w), N)
    smp = DeviceSeqSampler(inter, S, B, seed=3)
    nb = len(smp)
    for _ in smp:
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        for _ in smp:
            pass
    torch.cuda.synchronize()
    dts = (time.perf_counter() - t0) / 3
    model.train()

    def epoch_rate(sampler):
        coach = Coach(model, sampler, monitors=["LOSS"], kind="seq")
        coach.train_per_epoch(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e in range(3):
            coach.train_per_epoch(1 + e)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3
    dtc = epoch_rate(smp)
    dtf = epoch_rate(DeviceSeqSampler(inter, S, B, seed=3, fused=True))
    full = (U // B) * B + (U % B)
    return {"sampler_samples_per_sec": round(full / dts, 1), "sampler_ms_per_batch": round(dts / nb * 1e3, 4),
            "sampler_to_coach_samples_per_sec": round(full / dtf, 1), "sampler_to_coach_ms_per_step": round(dtf / nb * 1e3, 4),
            "sampler_to_coach_unfused_samples_per_sec": round(full / dtc, 1), "sampler_to_coach_unfused_ms_per_step": round(dtc / nb * 1e3, 4),
            "what": f"DeviceSeqSampler over {U} users ({int(ptr[-1])} interactions, {nb} batches of {B}; the last one short), 3 epochs each: the sampler "
                    "alone (one re_seq_train_sample launch per batch); Coach.train_per_epoch fed by the FUSED sampler (tickets, one ahead: every step's tail "
                    "launch samples and prepares the NEXT batch -- re_sasrec_step_stage_sample + re_next_prep -> one stage launch + one hipGraph replay per "
                    "step; the epoch's loss read once); and fed by the sampler's tensor batches (sampler launch per batch; the next batch prepared by the "
                    "tail launch likewise: `unfused`)"}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` started WITHOUT a launcher: start the N ranks here (one process per GPU over RCCL, rendezvous on 127.0.0.1)
    before this process has touched the GPU, and leave with their exit code -- a failed rank is a failed run, not a silent N = 1 line."""
    import socket
    import subprocess
    if torch.cuda.device_count() < n and "--rendezvous-only" not in argv:
        print(f"[bench] --gpus {n} asked for, {torch.cuda.device_count()} visible", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=dict(os.environ, RECBENCH_LAUNCHED="1")).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="skip the config-5 leg (100 M x 128 table: 154 GB of HBM)")
    ap.add_argument("--no-legs", action="store_true", help="skip the config-1 / config-3 / config-4 legs and the sampler rates")
    ap.add_argument("--no-extras", action="store_true", help="skip eval / gather legs (profiling runs)")
    ap.add_argument("--no-baselines", action="store_true", help="skip the aten-on-GPU baselines (kernel-trace runs: only the engine's kernels)")
    ap.add_argument("--encoder", default="fused", choices=("fused",), help="(the engine has one encoder: the HIP kernels; the torch comparator lives in tests/aten_sasrec.py)")
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false",
                    help="do not hand train_step_graph the next batch (default: an epoch loop has it; it is prepared by jobs of this step's tail launch)")
    ap.add_argument("--prefetch", dest="prefetch", action="store_true", help=argparse.SUPPRESS)
    ap.set_defaults(prefetch=True)
    ap.add_argument("--no-graph", action="store_true", help="launch the step's kernels one by one instead of replaying a hipGraph")
    ap.add_argument("--dp-outside-graph", action="store_true", help="--dp owner: issue the two collectives and the owner's launch behind the hipGraph replay "
                                                                    "instead of recording them into it")
    ap.add_argument("--dp", default="owner", choices=("owner", "allreduce"),
                    help="N > 1: how the replicas' step is synchronised.  owner (default): all-to-all of the gradient arenas, Adam on each slice's owner, "
                         "all-gather of the parameters (recboard_amd/dp.py); allreduce: one all-reduce of the gradient arena, dense Adam everywhere")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="every rank joins the process group (RCCL; gloo where there is no GPU), all-gathers (rank, world) and prints its own JSON line; "
                         "no engine work -- the launcher / rendezvous path alone (tests/test_bench_launcher.py runs it with four ranks on the CPU)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))      # (no GPU call has been made in this process)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE = {world} ranks", file=sys.stderr)
        sys.exit(2)
    if args.rendezvous_only:
        import torch.distributed as dist
        rank = int(os.environ.get("RANK", "0"))
        gpu = torch.cuda.is_available()
        if gpu:
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl" if gpu else "gloo", rank=rank, world_size=world)
        t = torch.tensor([rank, dist.get_world_size()], device="cuda" if gpu else "cpu")
        seen = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(seen, t)
        print(json.dumps({"rendezvous": "ok", "rank": rank, "world_size": dist.get_world_size(), "backend": dist.get_backend(),
                          "ranks_seen": sorted(int(x[0]) for x in seen), "master": os.environ.get("MASTER_ADDR")}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist = None
    force_dist = os.environ.get("RECBENCH_FORCE_DIST") == "1"   # smoke-test the multi-rank code path (RCCL + graph + hook) on one GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        assert dist.get_world_size() == world == (args.gpus if not force_dist else world), (dist.get_world_size(), world, args.gpus)

    import bench_legs
    from recboard_amd import ops
    from recboard_amd.sasrec import SASRecEngine
    cfg = BEAUTY
    model = SASRecEngine(cfg["items"], cfg["S"], cfg["D"], cfg["L"], dropout_rate=cfg["p_drop"], loss="BCE",
                         lr=cfg["lr"], weight_decay=cfg["wd"], seed=1, encoder=args.encoder)
    host_batches = synth_batches(cfg, 8, seed=1 + rank)
    batches = []
    for seq, pos, neg in host_batches:
        t = tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg))
        batches.append(t + ((None if args.encoder == 'fused' else model.batch_aux(*t)),))

    hook = None
    if (world > 1 or force_dist) and args.dp == "owner":
        from recboard_amd.dp import OwnerAdam
        hook = OwnerAdam(model.arena.numel)   # two one-hop collectives per step; the Adam launch covers 1 / world of the arena
        hook.in_graph = not args.dp_outside_graph
    elif world > 1 or force_dist:
        def hook(garena):  # ONE collective per step: the whole gradient arena is a single bucket
            dist.all_reduce(garena, op=dist.ReduceOp.AVG)   # (RCCL averages in the collective: no separate scaling launch)

    use_graph = args.encoder == "fused" and not args.no_graph
    torch.cuda.synchronize()

    def step_eager(i):
        seq, pos, neg, aux = batches[i % len(batches)]
        return model.train_step(seq, pos, neg, aux, grad_hook=hook)

    def step_graph(i):   # RAW batch in: one preparation launch (mask, count, scatter rows, encoder plan, staging) + one graph replay
        seq, pos, neg, _ = batches[i % len(batches)]
        nxt = batches[(i + 1) % len(batches)][:3] if args.prefetch else None   # (the next batch: prepared during this step)
        return model.train_step_graph(seq, pos, neg, grad_hook=hook, next_batch=nxt)

    # Graph or eager launches: decided BEFORE any step that contains a collective runs.  Every rank captures and replays one step
    # with a no-op hook (no collective), the ranks agree with one all-reduce, and the step's effects are rolled back -- a rank whose
    # capture failed would otherwise skip a collective the others wait in.
    step = step_eager
    if use_graph:
        A = model.arena
        keep = [t.clone() for t in (A.data, A.m, A.v)] + [A.step]
        try:
            seq0, pos0, neg0, _ = batches[0]
            # (an OwnerAdam hook is recorded INTO the graph, collectives included: every rank captures it here, at the same point, and its
            #  warm-up run is a real exchange; a plain all-reduce hook stays outside the graph and is rehearsed as a no-op)
            model.train_step_graph(seq0, pos0, neg0, grad_hook=(None if hook is None else (hook if getattr(hook, "owns_adam", False) else (lambda g: None))),
                                   next_batch=(seq0, pos0, neg0) if args.prefetch else None)      # (the same captured copies the timed steps replay)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            if dist is None:
                raise
            print(f"[bench] rank {rank}: hipGraph capture failed ({type(e).__name__}: {e})", file=sys.stderr)
            use_graph = False
        for t, k in zip((A.data, A.m, A.v), keep[:3]):
            t.copy_(k)
        A.step = keep[3]
        if dist is not None:
            flag = torch.tensor([1 if use_graph else 0], device="cuda")
            lo, hi = flag.clone(), flag.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if int(lo) != int(hi):
                print(f"[bench] rank {rank}: the ranks disagree on hipGraph capture; aborting", file=sys.stderr)
                sys.exit(3)
        step = step_graph if use_graph else step_eager

    for i in range(args.warmup):
        step(i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if hasattr(model, "check_handover"):
        model.check_handover()          # (split long sequences: no hand-over time-out during the timed region)
    ms = dt / args.steps * 1e3
    value = world * cfg["B"] * args.steps / dt

    line = {
        "metric": "train samples/sec (SASRec d=64, B=512/GPU; full-catalog items scored/sec in items_scored_per_sec)",
        "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (storage, accumulation, row-wise math); linear maps of the step as 3 bf16 split products", "data": "synthetic (Beauty-shaped, SURVEY.md §8d C2), random-init weights",
        "config": {"workload": "SASRec d=64 L=2 maxlen=50 BCE dropout=0.5 Adam on Amazon2014Beauty_550_LOU shapes "
                               "(12101 items, 22363 users), B=512 per GPU",
                   "global_batch": world * cfg["B"], "seq_len": cfg["S"],
                   "launch": (("one stage launch (step scalars, weight fragments, the next batch's addresses) + one hipGraph replay per step; every step "
                               "prepares the NEXT raw batch in jobs of its tail launch (train_step_graph(next_batch=...), as an epoch loop calls it)")
                              if args.prefetch else "one batch-preparation launch + one hipGraph replay per step")
                   if use_graph else "eager (one launch per kernel)",
                   "timed_region": "raw (seq, pos, neg) in HBM -> batch preparation (mask, count, rows, plan) -> forward, backward, Adam (dropout 0.5 on): "
                                   "every timed step does all of it for one batch",
                   "parallelism": (f"dp{world} (replicated 3 MB table; per step: all-to-all of the gradient arenas, Adam on each slice's owner, all-gather of the "
                                   f"parameters: {getattr(hook, 'bytes_per_link_per_step', 0)} B per xGMI link per step)" if getattr(hook, "owns_adam", False)
                                   else f"dp{world} (replicated 3 MB table, one gradient-arena all-reduce per step)")},
        "world_size": (dist.get_world_size() if dist is not None else 1), "backend": (dist.get_backend() if dist is not None else None),
        "final_loss": round(float(loss), 5),
    }

    if dist is not None and world > 1 and not args.no_c5:
        # ---------------- BASELINE.json configs[4] as named: the 100 M-row table row-sharded over the ranks (every rank takes part).
        # This leg's collectives run inside a captured graph on N > 1 ranks for the first time on the driver's node: if it has not come
        # back after 5 minutes every rank gives it up -- rank 0 prints the line measured so far -- instead of hanging the run.
        import threading
        leg_done = threading.Event()

        def give_up():
            if not leg_done.wait(300.0):
                if rank == 0:
                    line["config5_sharded"] = {"skipped": "no result after 300 s: abandoned (the headline above was measured before this leg)"}
                    print(json.dumps(line), flush=True)
                os._exit(3)
        threading.Thread(target=give_up, daemon=True).start()
        c5s = bench_legs.config5_sharded(dist, rank, world, local)
        leg_done.set()
        bad = torch.tensor([1 if "skipped" in c5s else 0], device="cuda")
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad) and "skipped" not in c5s:
            c5s = {"skipped": "another rank failed in this leg"}
        line["config5_sharded"] = c5s
    if rank == 0 and not args.no_extras and world == 1:
        # ---------------- spread of the step time (median + p10 / p90): every step of a second run bracketed by events on
        # the stream the step runs on, read after the run (nothing synchronises inside it)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(201)]
        evs[0].record()
        for i in range(200):
            step(i)
            evs[i + 1].record()
        torch.cuda.synchronize()
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(200))
        line["step_ms_percentiles"] = {"p10": round(per[20], 4), "p50": round(per[100], 4), "p90": round(per[180], 4), "steps": 200,
                                       "how": "hipEvents between consecutive steps of a second run (device-side spacing of the steps)"}
    if rank == 0 and not args.no_extras:
        # ---------------- dominant kernel of the timed region: the per-block encoder backward (MFMA-bound, fp32)
        if args.encoder == "fused":
            seq, pos, neg, _ = batches[0]
            Bq, Sq, Dq, Lq = cfg["B"], cfg["S"], cfg["D"], cfg["L"]
            W = model._buffers(Bq, Sq)
            A = model.arena
            G = A.views(A.grad)
            bt, bg = model._block_tensors(), model._block_tensors(A.grad)
            lw, lb = model.params["lastLN.weight"].detach(), model.params["lastLN.bias"].detach()
            dU = torch.randn(Bq, Sq, Dq, device="cuda") * 1e-3
            dx = torch.empty_like(dU)
            pb = model.prepare_batch(seq, pos, neg)

            def run_bwd():   # same launches as in the step: tape of the last training step, same plan
                ops.sasrec_encoder_bwd(dU, seq, bt, lw, lb, Lq, cfg["p_drop"], model._step_seed(), W["tape"], bg,
                                       G["lastLN.weight"], G["lastLN.bias"], out=dx, ws=W["ws_bwd"], plan=pb.plan)
            model.train_step(seq, pos, neg, pb)           # leaves this batch's tape in W["tape"]
            model.arena.step -= 1                         # keep the seed the tape was produced with
            t_bwd = graph_time_ms(run_bwd)

            def run_fwd():
                ops.sasrec_embed_encoder_fwd(model.params["Item.embeddings.weight"].detach(), model.params["Position.weight"].detach(), seq,
                                             float(Dq ** 0.5), bt, lw, lb, Lq, cfg["p_drop"], model._step_seed(), need_tape=True, out=W["u"],
                                             tape=W["tape"], plan=pb.plan)
            t_fwd = graph_time_ms(run_fwd)
            t_prep = graph_time_ms(lambda: ops.sasrec_batch_prep(seq, pos, neg, blob=pb.blob))

            def run_step():   # what the training step launches: forward + criterion + backward per work item, weight gradients, reduction
                ops.sasrec_encoder_step(model.params["Item.embeddings.weight"].detach(), model.params["Position.weight"].detach(), pb.seq, pb.pos,
                                        pb.neg, float(Dq ** 0.5), bt, lw, lb, Lq, cfg["p_drop"], model._step_seed(), pb.plan, ops.LOSS_BCE,
                                        pb.count, W["u"], W["tape"], W["dU_rows"], W["g_rows"], W["keys"], W["ws_loss"],
                                        W["contrib"][:Bq * Sq].view(Bq, Sq, Dq), G["Position.weight"], bg, G["lastLN.weight"], G["lastLN.bias"],
                                        W["ws_bwd"])
            t_step = graph_time_ms(run_step)
            model.arena.step += 1
            hdr = pb.plan.view(torch.int32)[:8].cpu().numpy()
            n_items, n_tiles = int(hdr[0]), int(hdr[1])
            # algorithmic work: 62 kFLOP per token per block forward, x2 for the backward, over ALL B*S token slots
            # the reference computes (pads included); executed: per tile and block 8 + 16 products of [16 rows] x D x D (forward; backward
            # incl. the weight gradients) + the attention products
            fl_ref = 3 * 62e3 * Bq * Sq * Lq
            fl_exec = Lq * n_tiles * (24 * 2 * 16 * Dq * Dq + 6 * 2 * 16 * 16 * Dq)
            tf_exec = fl_exec / (t_step * 1e-3) / 1e12
            n_real = int((seq > 0).sum())
            n_w = sum(int(t.numel()) for t in bt) + 2 * Dq
            alg_bytes = n_real * (3 * (8 + 4 * Dq) + 4 * 4 * Dq) + 2 * 4 * n_w
            tile_mode = int(hdr[7]) == 1
            line["encoder_launch_us"] = {"batch_prep": round(t_prep * 1e3, 1),
                                         "encoder step (re_sasrec_encoder_step, all its launches)": round(t_step * 1e3, 1),
                                         "forward alone (enc_fwd_k, tape)": round(t_fwd * 1e3, 1),
                                         "backward alone (enc_bwd_k + enc_wgrad_k + enc_grad_reduce_k)": round(t_bwd * 1e3, 1),
                                         "how": "each call captured 20x into a hipGraph, replayed 10x (GPU time; an eager loop is CPU-launch-bound)"}
            names = ("enc_tile_prep_k", "enc_tile_step_k", "enc_step_k<64>", "enc_wgrad_k<64>", "enc_grad_reduce_k")
            line["roofline"] = {"kernel": "re_sasrec_encoder_step: enc_tile_prep_k (weight fragments) + enc_tile_step_k (forward + criterion + backward of one "
                                          "16-token tile per workgroup, all blocks) + enc_wgrad_k + enc_grad_reduce_k (the launch group of the C-ABI "
                                          "entry; the captured step runs enc_wgrad_k's jobs inside its tail launch, enc_tail_k)"
                                          + ("" if tile_mode else " [this plan took the workgroup-per-item kernel enc_step_k instead]"), "bound": "mfma",
                                "achieved": round(tf_exec, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                "frac": round(tf_exec / MFMA_F32_PEAK_TF, 4),
                                "xdl_pipe": {"executed_bf16_TFLOPs": round(3 * tf_exec, 1), "peak": 2500.0, "frac": round(3 * tf_exec / 2500.0, 4),
                                             "note": "the tile kernel runs every fp32 product as 3 bf16 products (hi.hi + hi.mid + mid.hi, fp32 accumulate) on the XDL "
                                                     "pipe; `achieved` prices the algorithmic fp32 FLOPs against the fp32 matrix peak"},
                                "traffic": pmc_traffic(*names, algorithmic_bytes=alg_bytes), "launch_ms": round(t_step, 4),
                                "work": f"executed (`achieved`, `frac`): per 16-token tile and block 8 + 16 products of [16] x D x D (forward; backward incl. the "
                                        f"weight gradients) + the attention products = {fl_exec:.3e} FLOP on {n_tiles} tiles of real tokens in {n_items} work items "
                                        f"(the reference's aten path also computes the {Bq * Sq - n_real} pad slots: {fl_ref:.3e} FLOP, not counted here).  "
                                        f"Algorithmic bytes of the launch group: {n_real} real rows x (3 x (8 + 4 D) gathered + 4 x 4 D gradient / output rows) + "
                                        f"the weights and their gradients = {alg_bytes / 1e6:.1f} MB; the activation tape (written by the forward, read by the "
                                        f"backward and the weight-gradient jobs) is what `wasted_traffic` counts on top"}
        # ---------------- full-catalog evaluation leg: every user x every item, seen-mask + top-50 fused
        U, N, D, K = cfg["users"], cfg["items"], cfg["D"], 50
        rng = np.random.default_rng(7)
        model.eval()
        eval_seq = torch.from_numpy(np.concatenate([b[0] for b in synth_batches(cfg, (U + cfg["B"] - 1) // cfg["B"], 99)])[:U]).cuda()
        seen = [np.unique(s[s > 0] - 1) for s in eval_seq.cpu().numpy()]
        sp = np.zeros(U + 1, np.int64)
        sp[1:] = np.cumsum([len(x) for x in seen])
        seen_ptr, seen_idx = torch.from_numpy(sp).cuda(), torch.from_numpy(np.concatenate(seen)).cuda()
        with torch.no_grad():
            q = torch.cat([model.encode(eval_seq[i:i + cfg["B"]])[0][:, -1, :] for i in range(0, U, cfg["B"])]).contiguous()
        items = model.params["Item.embeddings.weight"].detach()[1:]
        t_score = event_time_ms(lambda: ops.score_topk(q, items, seen_ptr, seen_idx, K), 20)
        flops = 2.0 * D * U * N
        tf = flops / (t_score * 1e-3) / 1e12
        # the same call on iid scores (random unit-scale queries and items): the bench's own state after a few hundred steps keeps every
        # user's best items among the popular ids, which the threshold filter likes -- the iid figure is the data-independent one
        gq = torch.Generator(device="cuda").manual_seed(11)
        q_iid = torch.randn(U, D, device="cuda", generator=gq)
        E_iid = torch.randn(N, D, device="cuda", generator=gq)
        t_iid = event_time_ms(lambda: ops.score_topk(q_iid, E_iid, seen_ptr, seen_idx, K), 20)
        tf_iid = flops / (t_iid * 1e-3) / 1e12
        line["items_scored_per_sec"] = round(U * N / (t_score * 1e-3), 1)
        line["roofline_score"] = {"kernel": "re_score_topk: score_front_k (query split + item split + starting thresholds, one launch), score_kernel_reg<64,28,28,split> (bf16 hi/mid products on the XDL "
                                            "pipe), score_topk_merge_x<64> (exact fp32 re-scoring + certificate), fallback pass",
                            "bound": "mfma", "achieved": round(tf, 2),
                            "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4),
                            "iid_scores": {"launch_ms": round(t_iid, 4), "achieved": round(tf_iid, 2), "frac": round(tf_iid / MFMA_F32_PEAK_TF, 4)},
                            "xdl_pipe": {"executed_bf16_TFLOPs": round(3 * tf, 1), "peak": 2500.0, "frac": round(3 * tf / 2500.0, 4),
                                         "note": "`achieved` is a THROUGHPUT ratio: algorithmic fp32 FLOPs (2*D per pair) per second against the fp32 "
                                                 "MFMA peak; the arithmetic that runs is 3 bf16 products per pair on the XDL pipe (this entry), the "
                                                 "returned values are the exact fp32 chains"},
                            "traffic": score_call_traffic(), "launch_ms": round(t_score, 4),
                            "work": f"2*D*B*N = {flops:.3e} FLOP per call (B={U}, N={N}, D={D}, K={K})",
                            "whole_call": "all launches of one re_score_topk call, item-table split included"}
        del q_iid, E_iid
        # ---------------- the same evaluation as the reference executes it (UniSRec/main.py:408-414: dense scores, scores[seen] = -1e23,
        # torch.topk), through ROCm aten on this GPU and through torch on the host cores -- baselines, reported beside the engine
        if args.no_baselines:
            args.no_cpu_baseline = True
        def aten_eval(qq, EE, sp_, si_, rows):
            sc = qq @ EE.t()
            sc[rows, si_] = -1e23
            return torch.topk(sc, K, dim=1)
        if not args.no_baselines:
            rows_all = torch.repeat_interleave(torch.arange(U, device="cuda"), seen_ptr[1:] - seen_ptr[:-1])
            t_aten = event_time_ms(lambda: aten_eval(q, items, seen_ptr, seen_idx, rows_all), 5, warmup=2)
            va, ia = aten_eval(q, items, seen_ptr, seen_idx, rows_all)
            ve, ie = ops.score_topk(q, items, seen_ptr, seen_idx, K)
            agree = float((ia == ie).float().mean())       # (aten's GEMM sums in another order: near-ties may swap)
            line["eval_baselines"] = {"aten_gpu": {"items_per_sec": round(U * N / (t_aten * 1e-3), 1), "ms": round(t_aten, 3),
                                                   "what": "torch (ROCm aten) on the same GPU: q @ E.T, masked fill, torch.topk(50), all users in one batch",
                                                   "topk_index_agreement_with_engine": round(agree, 6)}}
            del va, ia, ve, ie
        if not args.no_cpu_baseline:
            nsub = 2048
            qc, Ec = q[:nsub].cpu(), items.cpu()
            spc = seen_ptr[:nsub + 1].cpu()
            sic = seen_idx[:int(spc[-1])].cpu()
            rc_ = torch.repeat_interleave(torch.arange(nsub), spc[1:] - spc[:-1])
            torch.set_num_threads(min(os.cpu_count() or 1, 16))

            def cpu_eval():
                sc = qc @ Ec.t()
                sc[rc_, sic] = -1e23
                return torch.topk(sc, K, dim=1)
            cpu_eval()
            t0 = time.time()
            nrep = 0
            while time.time() - t0 < 3.0:
                cpu_eval(); nrep += 1
            dt = (time.time() - t0) / nrep
            line["eval_baselines"]["torch_cpu"] = {"items_per_sec": round(nsub * N / dt, 1), "cores": torch.get_num_threads(),
                                                   "sample": f"{nsub} users x {N} items, {nrep} repetitions, {dt * 1e3:.1f} ms each"}
        # ---------------- embedding gather leg (HBM-bound): Beauty shape and an HBM-resident 4 GiB table
        idx_small = batches[0][0].reshape(-1)
        W_small = model.params["Item.embeddings.weight"].detach()
        t_g = event_time_ms(lambda: ops.gather_rows(W_small, idx_small), 50)
        R_big, n_big = 16 * 1024 * 1024, 4 * 1024 * 1024
        W_big = torch.empty((R_big, D), dtype=torch.float32, device="cuda").normal_()
        idx_big = torch.randint(0, R_big, (n_big,), device="cuda")
        out_big = None
        t_gb = event_time_ms(lambda: ops.gather_rows(W_big, idx_big), 20)
        bpr = 8 + 8 * D
        gbs_small = idx_small.numel() * bpr / (t_g * 1e-3) / 1e9
        gbs_big = n_big * bpr / (t_gb * 1e-3) / 1e9
        # this box's own streaming ceiling for the same bytes: the same kernel over SEQUENTIAL rows (a pure copy + the index stream) and
        # torch's device-to-device copy of the same 1 GiB (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured with a float4 copy)
        idx_seq = torch.arange(n_big, device="cuda")
        t_seq = event_time_ms(lambda: ops.gather_rows(W_big, idx_seq), 20)
        src_c = W_big[:n_big]
        dst_c = torch.empty_like(src_c)
        t_cp = event_time_ms(lambda: dst_c.copy_(src_c), 20)
        peak_meas = max(n_big * bpr / (t_seq * 1e-3) / 1e9, 2 * n_big * 4 * D / (t_cp * 1e-3) / 1e9)
        del dst_c, src_c, idx_seq
        line["roofline_gather"] = {"kernel": "gather_rows_vec4<16>", "bound": "hbm", "achieved": round(gbs_big, 1),
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs_big / HBM_PEAK_GBS, 4),
                                   "peak_measured": round(peak_meas, 1), "frac_of_peak_measured": round(gbs_big / peak_meas, 4),
                                   "peak_measured_how": f"best of this box, same run: the same kernel over sequential rows ({n_big * bpr / (t_seq * 1e-3) / 1e9:.0f} GB/s) "
                                                        f"and torch's device-to-device copy of 1 GiB ({2 * n_big * 4 * D / (t_cp * 1e-3) / 1e9:.0f} GB/s)",
                                   "traffic": pmc_traffic("gather_rows_vec4<16"), "launch_ms": round(t_gb, 4),
                                   "work": f"{bpr} B per looked-up row x {n_big} uniform-random rows of a {R_big}x{D} fp32 table (4 GiB, HBM-resident)",
                                   "beauty_shape": {"rows": int(idx_small.numel()), "launch_ms": round(t_g, 4), "GB/s": round(gbs_small, 1),
                                                    "note": "3.1 MB table is L2/Infinity-Cache resident: launch-latency bound"}}
        del W_big, idx_big, out_big
        # ---------------- the step as a USER runs it: the engine's Coach epoch loop over HOST batches (CoachForSASRec.train_per_epoch,
        # SASRec/main.py:242-258): per batch the H2D copies, the batch-preparation launch and the graph replay; one loss read per epoch
        if args.encoder == "fused":
            from recboard_amd.coach import Coach
            model.train()
            nb = 100
            hb = synth_batches(cfg, 10, seed=77)
            pipe = [{"User": torch.arange(cfg["B"]), "ISeq": torch.from_numpy(hb[i % 10][0]).pin_memory(),
                     "IPos": torch.from_numpy(hb[i % 10][1]).pin_memory(), "INeg": torch.from_numpy(hb[i % 10][2]).pin_memory()} for i in range(nb)]
            coach = Coach(model, pipe, monitors=["LOSS"], kind="seq")
            coach.train_per_epoch(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            coach.train_per_epoch(1)
            torch.cuda.synchronize()
            dtc = time.perf_counter() - t0
            line["coach_loop"] = {"samples_per_sec": round(nb * cfg["B"] / dtc, 1), "ms_per_step": round(dtc / nb * 1e3, 4),
                                  "what": f"Coach.train_per_epoch over {nb} HOST batches (pinned): one packed H2D copy per batch on a copy stream, several "
                                          "batches ahead; per step one stage launch + one graph replay (the next batch is prepared by the step's tail "
                                          "launch); the epoch's mean loss read once at the end"}
        if world == 1 and args.encoder == "fused" and not args.no_legs:
            line["sampler"] = sampler_rates(cfg, model)
            try:
                line["fp32_exact"] = fp32_exact_step(cfg)
            except Exception as e:  # noqa: BLE001
                line["fp32_exact"] = {"skipped": f"{type(e).__name__}: {e}"}
            try:
                line["roofline_large_batch"] = large_batch_roofline(cfg, B=4096)
                line["roofline_large_batch"]["B_8192"] = large_batch_roofline(cfg, B=8192)
                line["roofline_large_batch"]["B_2048"] = large_batch_roofline(cfg, B=2048)
            except Exception as e:  # noqa: BLE001  (a leg of its own: the headline stands without it)
                line["roofline_large_batch"] = {"skipped": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_c5:
            line["config5"] = bench_legs.run_child("config5")
        if world == 1 and not args.no_legs:
            for leg in ("config1", "config3", "config4"):
                line[leg] = bench_legs.run_child(leg, timeout=300)
        if not args.no_baselines:
            line["train_baseline_aten_gpu"] = aten_train_baseline(cfg, batches)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, host_batches)
    if rank == 0:
        # what a reader with a truncated line must still see comes first: the contract's keys, then the second headline and the rooflines
        first = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "items_scored_per_sec", "roofline", "roofline_score", "roofline_gather", "cpu_baseline", "fp32_exact", "config")
        line = {**{k: line[k] for k in first if k in line}, **{k: v for k, v in line.items() if k not in first}}
        for k in ("roofline", "roofline_score", "roofline_gather", "cpu_baseline"):      # inside them: the numbers in front of the prose
            if isinstance(line.get(k), dict):
                d = line[k]
                head = ("bound", "achieved", "peak", "unit", "frac", "value", "cores", "kind")
                line[k] = {**{q: d[q] for q in head if q in d}, **{q: v for q, v in d.items() if q not in head and not isinstance(v, str)},
                           **{q: v for q, v in d.items() if q not in head and isinstance(v, str)}}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()            # the other ranks wait for rank 0's extra legs instead of tearing the communicator down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
