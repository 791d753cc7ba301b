"""GPU: BASELINE.json configs[4] at its REAL size -- SASRec d = 128 on the synthetic 100 000 001 x 128 item table (51 GB + two Adam moment
tables = 154 GB of HBM; skipped where less is free) -- one training step of SASRecLargeTableEngine against the CPU oracle.

The table comes from the counter-based generator (recboard_amd/large.py: counter_normal_rows: a row is a pure function of (seed, row)), so the
rows a batch touches can be rebuilt on the host without the table: the oracle (oracle/sasrec.py fit + oracle/adam.py) runs on the COMPACTED
sub-table of the ~10 k touched rows with the ids remapped.  Checked: the loss; every dense parameter's gradient at 1e-4 of its largest entry;
the touched rows after the row-sparse Adam update; 10^6 sampled untouched rows bit-identical to their initial values with zero moments."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_one_step_on_the_100m_item_table_matches_the_oracle_on_the_touched_rows():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from bench_legs import c5_batches
    from oracle import adam as oadam, sasrec as osas
    from recboard_amd import ops
    from recboard_amd.large import SASRecLargeTableEngine, counter_normal_rows
    N, D, B, S, L = int(os.environ.get("RECTEST_C5_ITEMS", 100_000_000)), 128, 512, 50, 2
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    need = 3 * (N + 1) * D * 4 + (10 << 30)
    if free < need:
        pytest.skip(f"needs {need / 1e9:.0f} GB of HBM, {free / 1e9:.0f} GB free")
    lr, wd, seed, std = 1e-3, 1e-6, 1, 0.02
    eng = SASRecLargeTableEngine(N, S, D, L, dropout_rate=0.0, loss="BCE", lr=lr, weight_decay=wd, seed=seed, table_std=std, table_init="counter")
    seq, pos, neg = c5_batches(np.random.default_rng(5), 1, N, B, S)[0]            # Zipf(1.05) items, uniform negatives
    live = seq > 0
    rows = torch.unique(torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), seq[live], pos[live] + 1, neg[live] + 1]))      # touched table rows (+ the pad row)
    assert rows[0] == 0 and 5000 < rows.numel() < 3 * B * S
    E0 = eng.E[rows].cpu()
    ref0 = counter_normal_rows(rows, D, seed, std, "cuda").cpu()
    ref0[0] = 0.0                                                                # (row 0 = padding)
    assert torch.equal(E0, ref0), "the table's rows are not the counter generator's"
    dense0 = {k: v.detach().cpu().clone() for k, v in eng.params.items()}
    pb = eng.prepare_batch(seq, pos, neg)
    loss = float(eng.train_step(seq, pos, neg, aux=pb))
    eng.check_handover()
    # ---- the oracle on the compacted sub-table
    remap = lambda t: torch.searchsorted(rows, t)                                # noqa: E731  (table row -> row of the sub-table)
    seq_c = torch.where(live, remap(seq), torch.zeros_like(seq)).cpu()
    pos_c = torch.where(live, remap(pos + 1) - 1, torch.zeros_like(pos)).cpu()
    neg_c = torch.where(live, remap(neg + 1) - 1, torch.zeros_like(neg)).cpu()
    P = {k: v.clone().requires_grad_(True) for k, v in dense0.items()}
    P["Item.embeddings.weight"] = E0.clone().requires_grad_(True)
    tape = eng._buffers(B, S)["tape"]
    gate_report = {l: {} for l in range(L)}
    gates = {l: ((ops.sasrec_tape_array(tape, pb.plan, B, S, D, L, "HR", l) > 0).cpu(), 2e-5, gate_report[l]) for l in range(L)}
    ref = osas.fit(P, seq_c, pos_c, neg_c, "BCE", L, gates=gates)
    ref.backward()
    for l, rep in gate_report.items():
        # the borrowed gates, counted: fewer than 1e-4 of the real rows' pre-activations lie inside the window, and OUTSIDE it the engine's gate
        # is the oracle's sign exactly -- a broken gate cannot hide in the window
        assert rep["total"] > 0 and rep["window"] < 1e-4 * rep["total"], (l, rep)
        assert rep["mismatch_outside"] == 0, (l, rep)
    assert abs(loss - ref.item()) <= 2e-5 * abs(ref.item()), (loss, ref.item())
    Gv = eng.arena.views(eng.arena.grad)
    for k in dense0:                                                             # every dense gradient, entry by entry
        r = P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])
        err = (Gv[k].cpu() - r).abs().max().item()
        assert err <= 1e-4 * r.abs().max().item() + 1e-7, (k, err, r.abs().max().item())
    # the touched rows after the row-sparse Adam (first step from zero moments; coupled L2 on the touched rows)
    gE = P["Item.embeddings.weight"].grad.numpy()
    W, m, v = E0.numpy().copy(), np.zeros_like(gE), np.zeros_like(gE)
    idx = np.arange(1, rows.numel())
    oadam.sparse_adam_rows(W, m, v, idx, gE[1:], 1, lr, wd=wd, padding_idx=0)
    E1, m1, v1 = eng.E[rows].cpu().numpy(), eng.Em[rows].cpu().numpy(), eng.Ev[rows].cpu().numpy()
    assert np.array_equal(E1[0], np.zeros(D, np.float32))
    # (the first Adam step is lr * g / (|g| + eps): where |g| is far above eps the step is lr * sign(g) whatever the last bits of g)
    g_tot = gE + wd * E0.numpy()
    firm = np.abs(g_tot) > 1e-6
    assert firm[1:].mean() > 0.9
    assert np.abs(E1 - W)[firm].max() <= 5e-3 * lr, float(np.abs(E1 - W)[firm].max())
    assert np.abs(E1 - W).max() <= 1.01 * lr
    np.testing.assert_allclose(m1[1:], m[1:], rtol=2e-3, atol=1e-9)
    np.testing.assert_allclose(v1[1:], v[1:], rtol=4e-3, atol=1e-16)
    # ---- untouched rows: a million of them, bit-identical to their initial values, moments zero
    g = torch.Generator(device="cuda").manual_seed(11)
    cand = torch.randint(1, N + 1, (1_000_000,), device="cuda", generator=g)
    cand = cand[~torch.isin(cand, rows)]
    for c0 in range(0, cand.numel(), 1 << 17):
        c = cand[c0:c0 + (1 << 17)]
        assert torch.equal(eng.E[c], counter_normal_rows(c, D, seed, std, "cuda"))
        assert not eng.Em[c].any() and not eng.Ev[c].any()
