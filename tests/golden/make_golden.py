#!/usr/bin/env python3
"""Generate golden vectors by importing the REFERENCE model classes.

Runs only in the development container (needs /root/reference, read-only).  The
reference's ``<Model>/main.py`` files are imported in place with the stand-in
``freerec`` of ``_freerec_standin.py`` on ``sys.modules``; nothing of the
reference's source enters this repository -- only inputs/outputs (``*.npz``).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Fixtures (all fp32, seed fixed, dropout 0 so train-mode forward is deterministic):
  sasrec_bce.npz / sasrec_bpr.npz / sasrec_ce.npz / sasrec_bce_d128.npz (embedding_dim = 128: BASELINE configs[4]) : SASRec/main.py  fit loss, every param grad,
        recommend_from_full scores, encode() output, masked top-K (Coach.evaluate contract,
        UniSRec/main.py:400-447)
  mfbpr.npz    : MF-BPR/main.py   fit loss + table grads + full scores
  lightgcn.npz : LightGCN/main.py encode/fit (rec_loss, emb_loss) + grads + full scores
  deepfm.npz   : DeepFM/main.py   logits, loss, grads (train-mode BN), eval sigmoid scores
  dcn.npz      : DCN/main.py      logits, loss, grads (train-mode BN), eval sigmoid scores
  gru4rec_{bce,bpr}.npz : GRU4Rec/main.py fit loss + every gradient + full scores (dropouts 0)
  sgl.npz      : SGL/main.py      fit (rec_loss, emb_loss, ssl_loss) on two edge-dropout subgraphs (the uniform draws recorded) + grads + scores
  jgcf.npz     : JGCF/main.py     fit (rec_loss, emb_loss) + table gradients + [low | mid] tables + full scores
  gcn.npz      : GCN/main.py      fit rec_loss + every gradient + propagated tables + full scores
  stamp_{bce,ce}.npz, narm.npz, fmlprec_bpr.npz, bsarec_ce.npz : STAMP / NARM / FMLP-Rec / BSARec main.py  fit loss + every gradient + full scores (dropouts 0)
  bert4rec.npz : BERT4Rec/main.py fit loss (the mask draw recorded) + every gradient + full scores (dropout 0)
  ngcf.npz     : NGCF/main.py     fit (rec_loss, emb_loss) + every gradient + full scores on D^-1 (A + I)
  simgcl.npz   : SimGCL/main.py   fit (rec_loss, emb_loss, ssl_loss at eps = 0: the noise is torch.rand_like) + grads + full scores
  pool.npz     : SASRec / MF-BPR / LightGCN main.py  recommend_from_pool scores [B, 21] (`python make_golden.py pool`, from the other fixtures' states)
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _freerec_standin as standin  # noqa: E402

REF = "/root/reference"


def import_ref(model_dir, modname, overrides):
    fr, mods = standin.build(argv_defaults=overrides)
    for k in list(sys.modules):
        if k == "freerec" or k.startswith("freerec."):
            del sys.modules[k]
    sys.modules.update(mods)
    argv = sys.argv
    sys.argv = ["main.py"]
    try:
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, model_dir, "main.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return fr, mod


def sd_np(model):
    return {"param/" + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()
            if not v.is_sparse and v.layout == torch.strided}


def grads_np(model):
    return {"grad/" + k: (p.grad.detach().cpu().numpy().copy() if p.grad is not None
                          else np.zeros(tuple(p.shape), np.float32))
            for k, p in model.named_parameters()}


def make_seqs(g, B, S, N, min_len=1):
    """Left-padded item sequences (ids +1, 0 = pad), IPos/INeg 0-based, 0 on pads.
    Row contract: HSTU/sampler.py:54-62 + SASRec/main.py:143-157."""
    lens = torch.randint(min_len, S, (B,), generator=g)
    lens[0] = S  # one full-length row
    lens[1] = 1  # one single-item row
    seq = torch.zeros(B, S, dtype=torch.long)
    pos = torch.zeros(B, S, dtype=torch.long)
    neg = torch.zeros(B, S, dtype=torch.long)
    for b in range(B):
        L = int(lens[b])
        seq[b, S - L:] = torch.randint(0, N, (L,), generator=g) + 1
        pos[b, S - L:] = torch.randint(0, N, (L,), generator=g)
        neg[b, S - L:] = torch.randint(0, N, (L,), generator=g)
    return seq, pos, neg


def masked_topk(scores, seen_lists, K):
    """Coach.evaluate full-ranking contract (UniSRec/main.py:408-414): scores[seen] = -1e23, then top-K."""
    s = scores.clone()
    for b, items in enumerate(seen_lists):
        s[b, torch.as_tensor(items, dtype=torch.long)] = -1e23
    vals, idx = torch.topk(s, K, dim=1)
    return s, vals, idx


def ragged_np(lists):
    ptr = np.zeros(len(lists) + 1, np.int64)
    ptr[1:] = np.cumsum([len(x) for x in lists])
    flat = np.concatenate([np.asarray(x, np.int64) for x in lists]) if ptr[-1] else np.zeros(0, np.int64)
    return ptr, flat


def gen_sasrec(loss, embedding_dim=64):
    torch.manual_seed(1)
    fr, ref = import_ref("SASRec", f"ref_sasrec_{loss}_{embedding_dim}", dict(dropout_rate=0.0, loss=loss, embedding_dim=embedding_dim))
    N, B, S = 200, 8, ref.cfg.maxlen
    F = fr.data.fields.Field
    ds = fr.data.datasets.RecDataSet([F("USER", "USER", "ID", count=40), F("ITEM", "ITEM", "ID", count=N)])
    model = ref.SASRec(ds)
    # make biases / LN affine non-trivial so they are actually exercised
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias") or "LN" in n:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    seq, pos, neg = make_seqs(g, B, S, N)
    data = {model.ISeq: seq, model.IPos: pos, model.INeg: neg}
    out = {"in/seq": seq.numpy(), "in/pos": pos.numpy(), "in/neg": neg.numpy(),
           "cfg/N": np.int64(N), "cfg/D": np.int64(ref.cfg.embedding_dim),
           "cfg/num_blocks": np.int64(ref.cfg.num_blocks)}
    out.update(sd_np(model))
    model.train()
    losses = model(data)
    losses["rec_loss"].backward()
    out["out/rec_loss"] = losses["rec_loss"].detach().numpy()
    out.update(grads_np(model))
    model.eval()
    with torch.no_grad():
        userEmbds, itemEmbds = model.encode(data)
        scores = model(data, ranking="full")
    out["out/userEmbds"] = userEmbds.numpy()
    out["out/scores"] = scores.numpy()
    seen = [sorted(set((seq[b][seq[b] > 0] - 1).tolist())) for b in range(B)]
    sp, si = ragged_np(seen)
    _, vals, idx = masked_topk(scores, seen, 50)
    out.update({"in/seen_ptr": sp, "in/seen_idx": si, "out/topk_vals": vals.numpy(), "out/topk_idx": idx.numpy()})
    name = f"sasrec_{loss.lower()}" + ("" if embedding_dim == 64 else f"_d{embedding_dim}")
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: loss={float(out['out/rec_loss']):.6f}")


def gen_mfbpr():
    torch.manual_seed(1)
    fr, ref = import_ref("MF-BPR", "ref_mfbpr", {})
    U, N, B = 50, 80, 16
    F = fr.data.fields.Field
    ds = fr.data.datasets.RecDataSet([F("USER", "USER", "ID", count=U), F("ITEM", "ITEM", "ID", count=N)])
    model = ref.MF(ds)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():  # std=1e-4 init gives ~ln2 everywhere; spread values so the test is sharp
        for p in model.parameters():
            p.copy_(0.3 * torch.randn(p.shape, generator=g))
    users = torch.randint(0, U, (B, 1), generator=g)
    users[1] = users[0]  # duplicate user -> colliding scatter-add
    ipos = torch.randint(0, N, (B, 1), generator=g)
    ineg = torch.randint(0, N, (B, 1), generator=g)
    ineg[2] = ipos[3]  # same row hit through both tables' index lists
    data = {model.User: users, model.IPos: ipos, model.INeg: ineg}
    out = {"in/users": users.numpy(), "in/pos": ipos.numpy(), "in/neg": ineg.numpy()}
    out.update(sd_np(model))
    model.train()
    loss = model(data)["rec_loss"]
    loss.backward()
    out["out/rec_loss"] = loss.detach().numpy()
    out.update(grads_np(model))
    model.eval()
    with torch.no_grad():
        model.reset_ranking_buffers()
        scores = model(data, ranking="full")
    out["out/scores"] = scores.numpy()
    np.savez_compressed(os.path.join(HERE, "mfbpr.npz"), **out)
    print(f"mfbpr: loss={float(loss):.6f}")


def sym_norm_adj(U, N, edges):
    """Bipartite D^-1/2 A D^-1/2 as CSR (no self loops; NGCF/main.py:76-87 adds them explicitly,
    so the default `to_normalized_adj('sym')` does not)."""
    n = U + N
    A = torch.zeros(n, n)
    for u, i in edges:
        A[u, U + i] = 1.0
        A[U + i, u] = 1.0
    deg = A.sum(1)
    dinv = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
    A = dinv[:, None] * A * dinv[None, :]
    return A.to_sparse_csr()


def gen_lightgcn():
    torch.manual_seed(1)
    U, N, B = 30, 40, 16
    g = torch.Generator().manual_seed(5)
    edges = set()
    while len(edges) < 150:
        edges.add((int(torch.randint(0, U, (1,), generator=g)), int(torch.randint(0, N, (1,), generator=g))))
    adj = sym_norm_adj(U, N, sorted(edges))
    fr, ref = import_ref("LightGCN", "ref_lightgcn", {})
    F = fr.data.fields.Field
    ds = fr.data.datasets.RecDataSet([F("USER", "USER", "ID", count=U), F("ITEM", "ITEM", "ID", count=N)], adj=adj)
    model = ref.LightGCN(ds)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(0.3 * torch.randn(p.shape, generator=g))
    users = torch.randint(0, U, (B, 1), generator=g)
    ipos = torch.randint(0, N, (B, 1), generator=g)
    ineg = torch.randint(0, N, (B, 1), generator=g)
    data = {model.User: users, model.IPos: ipos, model.INeg: ineg}
    out = {"in/users": users.numpy(), "in/pos": ipos.numpy(), "in/neg": ineg.numpy(),
           "in/adj_crow": adj.crow_indices().numpy(), "in/adj_col": adj.col_indices().numpy(),
           "in/adj_val": adj.values().numpy(), "cfg/num_layers": np.int64(ref.cfg.num_layers),
           "cfg/weight_decay": np.float64(ref.cfg.weight_decay)}
    out.update({"param/User.embeddings.weight": model.User.embeddings.weight.detach().numpy().copy(),
                "param/Item.embeddings.weight": model.Item.embeddings.weight.detach().numpy().copy()})
    model.train()
    losses = model(data)
    # CoachForLightGCN.train_per_epoch (LightGCN/main.py:160)
    loss = losses["rec_loss"] + ref.cfg.weight_decay * losses["emb_loss"]
    loss.backward()
    out["out/rec_loss"] = losses["rec_loss"].detach().numpy()
    out["out/emb_loss"] = losses["emb_loss"].detach().numpy()
    out["out/loss"] = loss.detach().numpy()
    out.update(grads_np(model))
    model.eval()
    with torch.no_grad():
        ue, ie = model.encode()
        model.reset_ranking_buffers()
        scores = model(data, ranking="full")
    out["out/userEmbds"], out["out/itemEmbds"], out["out/scores"] = ue.numpy(), ie.numpy(), scores.numpy()
    np.savez_compressed(os.path.join(HERE, "lightgcn.npz"), **out)
    print(f"lightgcn: rec={float(losses['rec_loss']):.6f} emb={float(losses['emb_loss']):.6f}")


def gen_deepfm():
    torch.manual_seed(1)
    fr, ref = import_ref("DeepFM", "ref_deepfm",
                         dict(hidden_dims="32,24,16", batch_norm=True, hidden_dropout_rate=0.0))
    counts = [23, 41, 7, 7, 2, 3, 2, 9, 17, 29]
    F = fr.data.fields.Field
    fields = [F(f"F{i}", "EMBED", *(("USER", "ID") if i == 0 else ("ITEM", "ID") if i == 1 else ()), count=c)
              for i, c in enumerate(counts)]
    fields.append(F("LABEL", "LABEL"))
    ds = fr.data.datasets.RecDataSet(fields)
    model = ref.DeepFM(ds)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "embeddings" in n:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    B = 32
    data = {f: torch.randint(0, f.count, (B, 1), generator=g) for f in model.input_fields}
    labels = torch.randint(0, 2, (B, 1), generator=g)
    data[model.Label] = labels
    out = {"in/x": torch.cat([data[f] for f in model.input_fields], 1).numpy(), "in/labels": labels.numpy(),
           "cfg/counts": np.asarray(counts, np.int64)}
    out.update(sd_np(model))
    model.train()
    logits = model.encode(data)
    out["out/train_logits"] = logits.detach().numpy().copy()
    loss = model.criterion(logits, labels)
    loss.backward()
    out["out/rec_loss"] = loss.detach().numpy()
    out.update(grads_np(model))
    # running stats were updated by the train-mode forward; save post-step buffers for the eval pass
    out.update({"post/" + k: v.detach().numpy().copy() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k})
    # canonical per-field names (the nn.Module aliasing of Field objects yields duplicate state_dict keys)
    out = {k: v for k, v in out.items() if "input_fields" not in k and "/User." not in k and "/Item." not in k}
    for i, f in enumerate(model.input_fields):
        out[f"table/{i}"] = f.embeddings.weight.detach().numpy().copy()
        out[f"table_lr/{i}"] = f.embeddings_lr.weight.detach().numpy().copy()
        out[f"gtable/{i}"] = f.embeddings.weight.grad.numpy().copy()
        out[f"gtable_lr/{i}"] = f.embeddings_lr.weight.grad.numpy().copy()
    model.eval()
    with torch.no_grad():
        out["out/eval_scores"] = model(data, ranking="pool").numpy()
    np.savez_compressed(os.path.join(HERE, "deepfm.npz"), **out)
    print(f"deepfm: loss={float(loss):.6f}")


def gen_dcn():
    torch.manual_seed(1)
    fr, ref = import_ref("DCN", "ref_dcn", dict(hidden_dims="32,24", batch_norm=True, hidden_dropout_rate=0.0, num_layers=3, embedding_dim=10))
    counts = [23, 41, 7, 7, 2, 3, 2, 9, 17, 29]
    F = fr.data.fields.Field
    fields = [F(f"F{i}", "EMBED", *(("USER", "ID") if i == 0 else ("ITEM", "ID") if i == 1 else ()), count=c)
              for i, c in enumerate(counts)]
    fields.append(F("LABEL", "LABEL"))
    ds = fr.data.datasets.RecDataSet(fields)
    model = ref.DCN(ds)
    g = torch.Generator().manual_seed(13)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "embeddings" in n:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    B = 32
    data = {f: torch.randint(0, f.count, (B, 1), generator=g) for f in model.input_fields}
    labels = torch.randint(0, 2, (B, 1), generator=g)
    data[model.Label] = labels
    out = {"in/x": torch.cat([data[f] for f in model.input_fields], 1).numpy(), "in/labels": labels.numpy(),
           "cfg/counts": np.asarray(counts, np.int64), "cfg/num_layers": np.int64(ref.cfg.num_layers)}
    out.update(sd_np(model))
    model.train()
    logits = model.encode(data)
    out["out/train_logits"] = logits.detach().numpy().copy()
    loss = model.criterion(logits, labels)
    loss.backward()
    out["out/rec_loss"] = loss.detach().numpy()
    out.update(grads_np(model))
    out.update({"post/" + k: v.detach().numpy().copy() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k})
    out = {k: v for k, v in out.items() if "input_fields" not in k and "/User." not in k and "/Item." not in k and "/fields." not in k}
    for i, f in enumerate(model.input_fields):
        out[f"table/{i}"] = f.embeddings.weight.detach().numpy().copy()
        out[f"gtable/{i}"] = f.embeddings.weight.grad.numpy().copy()
    model.eval()
    with torch.no_grad():
        out["out/eval_scores"] = model(data, ranking="pool").numpy()
    np.savez_compressed(os.path.join(HERE, "dcn.npz"), **out)
    print(f"dcn: loss={float(loss):.6f}")


def gen_simgcl():
    torch.manual_seed(1)
    U, N, B = 30, 40, 16
    g = torch.Generator().manual_seed(9)
    edges = set()
    while len(edges) < 150:
        edges.add((int(torch.randint(0, U, (1,), generator=g)), int(torch.randint(0, N, (1,), generator=g))))
    adj = sym_norm_adj(U, N, sorted(edges))
    fr, ref = import_ref("SimGCL", "ref_simgcl", dict(eps=0.0))      # (the perturbation is torch.rand_like: eps = 0 makes fit deterministic)
    F = fr.data.fields.Field
    ds = fr.data.datasets.RecDataSet([F("USER", "USER", "ID", count=U), F("ITEM", "ITEM", "ID", count=N)], adj=adj)
    model = ref.SimGCL(ds)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(0.3 * torch.randn(p.shape, generator=g))
    users = torch.randint(0, U, (B, 1), generator=g)
    ipos = torch.randint(0, N, (B, 1), generator=g)
    ineg = torch.randint(0, N, (B, 1), generator=g)
    data = {model.User: users, model.IPos: ipos, model.INeg: ineg}
    out = {"in/users": users.numpy(), "in/pos": ipos.numpy(), "in/neg": ineg.numpy(),
           "in/adj_crow": adj.crow_indices().numpy(), "in/adj_col": adj.col_indices().numpy(), "in/adj_val": adj.values().numpy(),
           "cfg/num_layers": np.int64(ref.cfg.num_layers), "cfg/temperature": np.float64(ref.cfg.temperature),
           "cfg/weight_decay": np.float64(ref.cfg.weight_decay), "cfg/lambda": np.float64(getattr(ref.cfg, "lambda_", getattr(ref.cfg, "ssl_weight", 0.0)) or 0.0)}
    out.update({"param/User.embeddings.weight": model.User.embeddings.weight.detach().numpy().copy(),
                "param/Item.embeddings.weight": model.Item.embeddings.weight.detach().numpy().copy()})
    model.train()
    losses = model(data)
    loss = losses["rec_loss"] + losses["emb_loss"] + losses["ssl_loss"]     # (unit weights: every term's gradient is exercised)
    loss.backward()
    for k in ("rec_loss", "emb_loss", "ssl_loss"):
        out["out/" + k] = losses[k].detach().numpy()
    out.update(grads_np(model))
    model.eval()
    with torch.no_grad():
        ue, ie = model.encode()
        model.reset_ranking_buffers()
        scores = model(data, ranking="full")
    out["out/userEmbds"], out["out/itemEmbds"], out["out/scores"] = ue.numpy(), ie.numpy(), scores.numpy()
    np.savez_compressed(os.path.join(HERE, "simgcl.npz"), **out)
    print("simgcl: " + " ".join(f"{k}={float(v):.6f}" for k, v in losses.items()))


def gen_ngcf():
    torch.manual_seed(1)
    U, N, B = 30, 40, 16
    g = torch.Generator().manual_seed(23)
    edges = set()
    while len(edges) < 150:
        edges.add((int(torch.randint(0, U, (1,), generator=g)), int(torch.randint(0, N, (1,), generator=g))))
    e = torch.tensor(sorted(edges), dtype=torch.long)
    ei = torch.stack((torch.cat((e[:, 0], e[:, 1] + U)), torch.cat((e[:, 1] + U, e[:, 0]))))     # both directions
    fr, ref = import_ref("NGCF", "ref_ngcf", dict(dropout_rate=0.0))
    F = fr.data.fields.Field
    ds = fr.data.datasets.RecDataSet([F("USER", "USER", "ID", count=U), F("ITEM", "ITEM", "ID", count=N)], edge_index=ei)
    model = ref.NGCF(ds)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif "embeddings" in n:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
    users = torch.randint(0, U, (B, 1), generator=g)
    ipos = torch.randint(0, N, (B, 1), generator=g)
    ineg = torch.randint(0, N, (B, 1), generator=g)
    data = {model.User: users, model.IPos: ipos, model.INeg: ineg}
    adj = model.Adj
    out = {"in/users": users.numpy(), "in/pos": ipos.numpy(), "in/neg": ineg.numpy(),
           "in/adj_crow": adj.crow_indices().numpy(), "in/adj_col": adj.col_indices().numpy(), "in/adj_val": adj.values().numpy(),
           "cfg/num_layers": np.int64(ref.cfg.num_layers)}
    out.update({k: v for k, v in sd_np(model).items() if "Adj" not in k})
    model.train()
    losses = model(data)
    (losses["rec_loss"] + losses["emb_loss"]).backward()
    out["out/rec_loss"], out["out/emb_loss"] = losses["rec_loss"].detach().numpy(), losses["emb_loss"].detach().numpy()
    out.update(grads_np(model))
    model.eval()
    with torch.no_grad():
        ue, ie = model.encode()
        model.reset_ranking_buffers()
        scores = model(data, ranking="full")
    out["out/userEmbds"], out["out/itemEmbds"], out["out/scores"] = ue.numpy(), ie.numpy(), scores.numpy()
    np.savez_compressed(os.path.join(HERE, "ngcf.npz"), **out)
    print(f"ngcf: rec={float(losses['rec_loss'].detach()):.6f} emb={float(losses['emb_loss'].detach()):.6f}")


def gen_gru4rec(loss):
    torch.manual_seed(1)
    fr, ref = import_ref("GRU4Rec", f"ref_gru4rec_{loss}", dict(loss=loss, emb_dropout_rate=0.0, hidden_dropout_rate=0.0, hidden_size=48,
                                                                embedding_dim=64, num_blocks=1))
    N, B, S = 150, 12, 20
    F = fr.data.fields.Field
    ds = fr.data.datasets.RecDataSet([F("USER", "USER", "ID", count=30), F("ITEM", "ITEM", "ID", count=N)])
    model = ref.GRU4Rec(ds)
    g = torch.Generator().manual_seed(17)
    with torch.no_grad():
        model.dense.bias.add_(0.1 * torch.randn(model.dense.bias.shape, generator=g))
    lens = torch.randint(1, S - 2, (B,), generator=g)          # (no row reaches S: shrink_pads drops the all-padding columns)
    seq = torch.zeros(B, S, dtype=torch.long)
    for b in range(B):
        seq[b, : int(lens[b])] = torch.randint(0, N, (int(lens[b]),), generator=g) + 1      # RIGHT-padded (GRU4Rec/main.py:93-96)
    pos = torch.randint(0, N, (B, 1), generator=g)
    neg = torch.randint(0, N, (B, 1), generator=g)
    data = {model.ISeq: seq, model.IPos: pos, model.INeg: neg}
    out = {"in/seq": seq.numpy(), "in/pos": pos.numpy(), "in/neg": neg.numpy(), "cfg/N": np.int64(N), "cfg/hidden": np.int64(48)}
    out.update(sd_np(model))
    model.train()
    losses = model(data)
    losses["rec_loss"].backward()
    out["out/rec_loss"] = losses["rec_loss"].detach().numpy()
    out.update(grads_np(model))
    model.eval()
    with torch.no_grad():
        out["out/scores"] = model(data, ranking="full").numpy()
    np.savez_compressed(os.path.join(HERE, f"gru4rec_{loss.lower()}.npz"), **out)
    print(f"gru4rec {loss}: loss={float(losses['rec_loss'].detach()):.6f}")


def gen_sgl():
    torch.manual_seed(1)
    U, N, B, rate = 30, 40, 16, 0.3
    g = torch.Generator().manual_seed(19)
    edges = set()
    while len(edges) < 150:
        edges.add((int(torch.randint(0, U, (1,), generator=g)), int(torch.randint(0, N, (1,), generator=g))))
    edges = sorted(edges)
    adj = sym_norm_adj(U, N, edges)
    u2i = torch.tensor(edges, dtype=torch.long).t().contiguous()            # [2, E]: (user, item)
    fr, ref = import_ref("SGL", "ref_sgl", dict(aug_type="ed", ssl_drop_rate=rate))
    F = fr.data.fields.Field
    ds = fr.data.datasets.RecDataSet([F("USER", "USER", "ID", count=U), F("ITEM", "ITEM", "ID", count=N)], adj=adj, u2i=u2i)
    model = ref.SGL(ds)
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(0.3 * torch.randn(p.shape, generator=g))
    users = torch.randint(0, U, (B, 1), generator=g)
    ipos = torch.randint(0, N, (B, 1), generator=g)
    ineg = torch.randint(0, N, (B, 1), generator=g)
    E = u2i.shape[1]
    torch.manual_seed(5)
    r1, r2 = torch.rand(E), torch.rand(E)                                   # what resample() will draw (SGL/main.py:98-100, 113-118)
    torch.manual_seed(5)
    model.resample()
    data = {model.User: users, model.IPos: ipos, model.INeg: ineg}
    out = {"in/users": users.numpy(), "in/pos": ipos.numpy(), "in/neg": ineg.numpy(), "in/edges": u2i.numpy(), "in/rnd1": r1.numpy(), "in/rnd2": r2.numpy(),
           "cfg/num_layers": np.int64(ref.cfg.num_layers), "cfg/temperature": np.float64(ref.cfg.temperature), "cfg/ssl_drop_rate": np.float64(rate),
           "out/sub_adj_dense": model.sAdjs[0].to_dense().numpy()}
    out.update({"param/User.embeddings.weight": model.User.embeddings.weight.detach().numpy().copy(),
                "param/Item.embeddings.weight": model.Item.embeddings.weight.detach().numpy().copy()})
    model.train()
    losses = model(data)
    (losses["rec_loss"] + losses["emb_loss"] + losses["ssl_loss"]).backward()
    for k in ("rec_loss", "emb_loss", "ssl_loss"):
        out["out/" + k] = losses[k].detach().numpy()
    out.update(grads_np(model))
    model.eval()
    with torch.no_grad():
        ue, ie = model.encode()
        model.reset_ranking_buffers()
        scores = model(data, ranking="full")
    out["out/userEmbds"], out["out/itemEmbds"], out["out/scores"] = ue.numpy(), ie.numpy(), scores.numpy()
    np.savez_compressed(os.path.join(HERE, "sgl.npz"), **out)
    print("sgl: " + " ".join(f"{k}={float(v.detach()):.6f}" for k, v in losses.items()))


def gen_jgcf():
    torch.manual_seed(1)
    U, N, B = 30, 40, 16
    g = torch.Generator().manual_seed(13)
    edges = set()
    while len(edges) < 150:
        edges.add((int(torch.randint(0, U, (1,), generator=g)), int(torch.randint(0, N, (1,), generator=g))))
    adj = sym_norm_adj(U, N, sorted(edges))
    sys.path.insert(0, os.path.join(REF, "JGCF"))                 # (JGCF/main.py imports its sibling modules.py)
    try:
        fr, ref = import_ref("JGCF", "ref_jgcf", dict(alpha=1.5, beta=0.5))      # (alpha != beta: the c1 term of the recurrence is live)
    finally:
        sys.path.pop(0)
        sys.modules.pop("modules", None)
    F = fr.data.fields.Field
    ds = fr.data.datasets.RecDataSet([F("USER", "USER", "ID", count=U), F("ITEM", "ITEM", "ID", count=N)], adj=adj)
    model = ref.JGCF(ds)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if p.requires_grad:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
    users = torch.randint(0, U, (B, 1), generator=g)
    ipos = torch.randint(0, N, (B, 1), generator=g)
    ineg = torch.randint(0, N, (B, 1), generator=g)
    data = {model.User: users, model.IPos: ipos, model.INeg: ineg}
    out = {"in/users": users.numpy(), "in/pos": ipos.numpy(), "in/neg": ineg.numpy(),
           "in/adj_crow": adj.crow_indices().numpy(), "in/adj_col": adj.col_indices().numpy(), "in/adj_val": adj.values().numpy(),
           "cfg/num_layers": np.int64(ref.cfg.num_layers), "cfg/alpha": np.float64(ref.cfg.alpha), "cfg/beta": np.float64(ref.cfg.beta),
           "cfg/scaling_factor": np.float64(ref.cfg.scaling_factor), "cfg/weight4mid": np.float64(ref.cfg.weight4mid)}
    out.update(sd_np(model))
    model.train()
    losses = model(data)
    (losses["rec_loss"] + losses["emb_loss"]).backward()
    for k in ("rec_loss", "emb_loss"):
        out["out/" + k] = losses[k].detach().numpy()
    out.update(grads_np(model))
    model.eval()
    with torch.no_grad():
        ue, ie = model.encode()
        model.reset_ranking_buffers()
        scores = model(data, ranking="full")
    out["out/userEmbds"], out["out/itemEmbds"], out["out/scores"] = ue.numpy(), ie.numpy(), scores.numpy()
    np.savez_compressed(os.path.join(HERE, "jgcf.npz"), **out)
    print("jgcf: " + " ".join(f"{k}={float(v):.6f}" for k, v in losses.items()))


def gen_gcn():
    torch.manual_seed(1)
    U, N, B = 30, 40, 16
    g = torch.Generator().manual_seed(21)
    edges = set()
    while len(edges) < 150:
        edges.add((int(torch.randint(0, U, (1,), generator=g)), int(torch.randint(0, N, (1,), generator=g))))
    adj = sym_norm_adj(U, N, sorted(edges))
    fr, ref = import_ref("GCN", "ref_gcn", dict())
    F = fr.data.fields.Field
    ds = fr.data.datasets.RecDataSet([F("USER", "USER", "ID", count=U), F("ITEM", "ITEM", "ID", count=N)], adj=adj)
    model = ref.GCN(ds)
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_((0.3 if "embeddings" in k else 0.2 if k.endswith("weight") else 0.1) * torch.randn(p.shape, generator=g))
    users = torch.randint(0, U, (B, 1), generator=g)
    ipos = torch.randint(0, N, (B, 1), generator=g)
    ineg = torch.randint(0, N, (B, 1), generator=g)
    data = {model.User: users, model.IPos: ipos, model.INeg: ineg}
    out = {"in/users": users.numpy(), "in/pos": ipos.numpy(), "in/neg": ineg.numpy(),
           "in/adj_crow": adj.crow_indices().numpy(), "in/adj_col": adj.col_indices().numpy(), "in/adj_val": adj.values().numpy(),
           "cfg/num_layers": np.int64(ref.cfg.num_layers)}
    out.update(sd_np(model))
    model.train()
    losses = model(data)
    losses["rec_loss"].backward()
    out["out/rec_loss"] = losses["rec_loss"].detach().numpy()
    out.update(grads_np(model))
    model.eval()
    with torch.no_grad():
        ue, ie = model.encode()
        model.reset_ranking_buffers()
        scores = model(data, ranking="full")
    out["out/userEmbds"], out["out/itemEmbds"], out["out/scores"] = ue.numpy(), ie.numpy(), scores.numpy()
    np.savez_compressed(os.path.join(HERE, "gcn.npz"), **out)
    print(f"gcn: rec_loss={float(losses['rec_loss'].detach()):.6f}")


def gen_last_item_model(model_dir, cls, outname, overrides, left_padded, with_modules=False, S=20, emb_scale=1.0):
    """STAMP / NARM / FMLP-Rec: one user state per sequence against the item table (IPos / INeg [B, 1])."""
    torch.manual_seed(1)
    if with_modules:
        sys.path.insert(0, os.path.join(REF, model_dir))
    try:
        fr, ref = import_ref(model_dir, "ref_" + outname, overrides)
    finally:
        if with_modules:
            sys.path.pop(0)
            sys.modules.pop("modules", None)
    N, B = 150, 12
    F = fr.data.fields.Field
    ds = fr.data.datasets.RecDataSet([F("USER", "USER", "ID", count=30), F("ITEM", "ITEM", "ID", count=N)])
    model = getattr(ref, cls)(ds)
    g = torch.Generator().manual_seed(31)
    with torch.no_grad():                                       # biases / LayerNorm affine / STAMP's ba off their initial values
        for k, p in model.named_parameters():
            if k.endswith("bias") or k == "ba" or "Norm" in k:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
            if k == "Item.embeddings.weight":                   # (STAMP's std = 0.002 tables give logits ~ 1e-5: scaled up to exercise the maths)
                p.mul_(emb_scale)
    lens = torch.randint(1, S - 2, (B,), generator=g)
    seq = torch.zeros(B, S, dtype=torch.long)
    for b in range(B):
        L = int(lens[b])
        it = torch.randint(0, N, (L,), generator=g) + 1
        if left_padded:
            seq[b, S - L:] = it
        else:
            seq[b, :L] = it
    pos = torch.randint(0, N, (B, 1), generator=g)
    neg = torch.randint(0, N, (B, 1), generator=g)
    data = {model.ISeq: seq, model.IPos: pos, model.INeg: neg}
    out = {"in/seq": seq.numpy(), "in/pos": pos.numpy(), "in/neg": neg.numpy(), "cfg/N": np.int64(N), "cfg/maxlen": np.int64(S)}
    out.update(sd_np(model))
    model.train()
    losses = model(data)
    losses["rec_loss"].backward()
    out["out/rec_loss"] = losses["rec_loss"].detach().numpy()
    out.update(grads_np(model))
    model.eval()
    with torch.no_grad():
        out["out/scores"] = model(data, ranking="full").numpy()
    np.savez_compressed(os.path.join(HERE, outname + ".npz"), **out)
    print(f"{outname}: loss={float(losses['rec_loss'].detach()):.6f}")


def gen_bert4rec():
    torch.manual_seed(1)
    S = 20
    fr, ref = import_ref("BERT4Rec", "ref_bert4rec", dict(dropout_rate=0.0, embedding_dim=64, num_heads=4, num_blocks=2, maxlen=S, mask_ratio=0.3))
    N, B = 150, 12
    F = fr.data.fields.Field
    ds = fr.data.datasets.RecDataSet([F("USER", "USER", "ID", count=30), F("ITEM", "ITEM", "ID", count=N)])
    model = ref.BERT4Rec(ds)
    g = torch.Generator().manual_seed(23)
    with torch.no_grad():                                       # biases / LayerNorm affine off their initial 0 / 1
        for k, p in model.named_parameters():
            if k.endswith("bias") or "norm" in k:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    lens = torch.randint(2, S + 1, (B,), generator=g)
    lens[0] = S
    seq = torch.zeros(B, S, dtype=torch.long)
    for b in range(B):
        seq[b, S - int(lens[b]):] = torch.randint(0, N, (int(lens[b]),), generator=g) + 2      # LEFT-padded, ids + NUM_PADS
    torch.manual_seed(5)
    rnds = torch.rand(seq.size())                               # what random_mask will draw (BERT4Rec/main.py:157)
    out = {"in/seq": seq.numpy(), "in/rnds": rnds.numpy(), "cfg/N": np.int64(N), "cfg/maxlen": np.int64(S)}
    out.update(sd_np(model))
    model.train()
    torch.manual_seed(5)
    data = {model.ISeq: seq.clone()}
    losses = model(data)
    assert torch.equal(data[model.ISeq] == 1, (rnds < 0.3) & (seq != 0))
    losses["rec_loss"].backward()
    out["out/rec_loss"] = losses["rec_loss"].detach().numpy()
    out["out/n_masked"] = np.int64(int((data[model.ISeq] == 1).sum()))
    out.update(grads_np(model))
    model.eval()
    seq_eval = torch.cat((seq[:, 1:], torch.ones(B, 1, dtype=torch.long)), 1)     # lpad_(maxlen - 1) + rpad_(maxlen, MASKING_VALUE)
    out["in/seq_eval"] = seq_eval.numpy()
    with torch.no_grad():
        out["out/states_eval"] = model.encode({model.ISeq: seq_eval}).numpy()
        out["out/scores"] = model({model.ISeq: seq_eval}, ranking="full").numpy()
    np.savez_compressed(os.path.join(HERE, "bert4rec.npz"), **out)
    print(f"bert4rec: loss={float(losses['rec_loss'].detach()):.6f} masked={int(out['out/n_masked'])}")


def gen_pool():
    """pool.npz: `recommend_from_pool` of the reference's SASRec / MF / LightGCN (SASRec/main.py:230-236, MF-BPR/main.py:106-109,
    LightGCN/main.py:122-125) under the states of sasrec_bce.npz / mfbpr.npz / lightgcn.npz (loaded, not regenerated): per evaluation row a
    pool of 1 + 20 item ids (the target first, as the evaluation pipes hand it over), with planted duplicates of the target (ties)."""
    out = {}
    g = torch.Generator().manual_seed(11)
    # SASRec
    z = np.load(os.path.join(HERE, "sasrec_bce.npz"))
    fr, ref = import_ref("SASRec", "ref_sasrec_pool", dict(dropout_rate=0.0, loss="BCE", embedding_dim=64))
    F = fr.data.fields.Field
    N = int(z["cfg/N"])
    model = ref.SASRec(fr.data.datasets.RecDataSet([F("USER", "USER", "ID", count=40), F("ITEM", "ITEM", "ID", count=N)]))
    model.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}, strict=True)
    seq = torch.from_numpy(z["in/seq"])
    pool = torch.randint(0, N, (seq.shape[0], 21), generator=g)
    pool[0, 5] = pool[0, 0]; pool[3, 20] = pool[3, 0]                  # the target again further down: a tie the target must win
    model.eval()
    with torch.no_grad():
        sc = model({model.ISeq: seq, model.IUnseen: pool}, ranking="pool")
    out["sasrec/pool"], out["sasrec/scores"] = pool.numpy(), sc.numpy()
    # MF-BPR
    z = np.load(os.path.join(HERE, "mfbpr.npz"))
    fr, ref = import_ref("MF-BPR", "ref_mfbpr_pool", {})
    F = fr.data.fields.Field
    U, N = z["param/User.embeddings.weight"].shape[0], z["param/Item.embeddings.weight"].shape[0]
    model = ref.MF(fr.data.datasets.RecDataSet([F("USER", "USER", "ID", count=U), F("ITEM", "ITEM", "ID", count=N)]))
    model.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}, strict=True)
    users = torch.from_numpy(z["in/users"])
    pool = torch.randint(0, N, (users.shape[0], 21), generator=g)
    model.eval()
    with torch.no_grad():
        model.reset_ranking_buffers()
        sc = model({model.User: users, model.IUnseen: pool}, ranking="pool")
    out["mfbpr/pool"], out["mfbpr/scores"] = pool.numpy(), sc.numpy()
    # LightGCN
    z = np.load(os.path.join(HERE, "lightgcn.npz"))
    U, N = z["param/User.embeddings.weight"].shape[0], z["param/Item.embeddings.weight"].shape[0]
    adj = torch.sparse_csr_tensor(torch.from_numpy(z["in/adj_crow"]), torch.from_numpy(z["in/adj_col"]), torch.from_numpy(z["in/adj_val"]), size=(U + N, U + N))
    fr, ref = import_ref("LightGCN", "ref_lightgcn_pool", {})
    F = fr.data.fields.Field
    model = ref.LightGCN(fr.data.datasets.RecDataSet([F("USER", "USER", "ID", count=U), F("ITEM", "ITEM", "ID", count=N)], adj=adj))
    with torch.no_grad():
        model.User.embeddings.weight.copy_(torch.from_numpy(z["param/User.embeddings.weight"]))
        model.Item.embeddings.weight.copy_(torch.from_numpy(z["param/Item.embeddings.weight"]))
    users = torch.from_numpy(z["in/users"])
    pool = torch.randint(0, N, (users.shape[0], 21), generator=g)
    model.eval()
    with torch.no_grad():
        model.reset_ranking_buffers()
        sc = model({model.User: users, model.IUnseen: pool}, ranking="pool")
    out["lightgcn/pool"], out["lightgcn/scores"] = pool.numpy(), sc.numpy()
    np.savez_compressed(os.path.join(HERE, "pool.npz"), **out)
    print("pool:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    torch.set_num_threads(1)
    if sys.argv[1:] == ["pool"]:          # (the pool fixture alone: it reads the other fixtures' states)
        gen_pool()
        sys.exit(0)
    for loss in ("BCE", "BPR", "CE"):
        gen_sasrec(loss)
    gen_sasrec("BCE", embedding_dim=128)
    gen_mfbpr()
    gen_lightgcn()
    gen_deepfm()
    gen_dcn()
    gen_simgcl()
    gen_ngcf()
    for loss in ("BCE", "BPR"):
        gen_gru4rec(loss)
    gen_sgl()
    gen_jgcf()
    gen_gcn()
    gen_last_item_model("STAMP", "STAMP", "stamp_bce", dict(loss="BCE", embedding_dim=64, hidden_size=64), True, emb_scale=200.0)
    gen_last_item_model("STAMP", "STAMP", "stamp_ce", dict(loss="CE", embedding_dim=64, hidden_size=64), True, emb_scale=200.0)
    gen_last_item_model("NARM", "NARM", "narm", dict(embedding_dim=64, hidden_size=48, num_blocks=1, emb_dropout_rate=0.0, hidden_dropout_rate=0.0,
                                                     ct_dropout_rate=0.0), False)
    gen_last_item_model("FMLP-Rec", "FMLPRec", "fmlprec_bpr", dict(loss="BPR", embedding_dim=64, num_blocks=2, hidden_dropout_rate=0.0, maxlen=20),
                        True, with_modules=True, emb_scale=10.0)
    gen_last_item_model("BSARec", "BSARec", "bsarec_ce", dict(loss="CE", embedding_dim=64, num_heads=2, num_blocks=2, hidden_dropout_rate=0.0,
                                                              attn_dropout_rate=0.0, maxlen=20, c=5, alpha=0.7), True, with_modules=True, emb_scale=10.0)
    gen_bert4rec()
    gen_pool()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
