"""Per-tensor numbers of the adoption probe (recboard_amd/bridge.py) for the example scripts: the script's own step gradient against the
engine's, tensor by tensor.  `python scripts/probe_diag.py [deepfm|deepfm_big]`."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_bridge_host import deepfm_dataset, load_deepfm_golden  # noqa: E402
from test_freerec_compat import G, import_script  # noqa: E402

from recboard_amd import bridge  # noqa: E402


def main(which):
    z = np.load(os.path.join(G, "deepfm.npz"))
    big = which == "deepfm_big"
    mod = import_script(os.path.join(ROOT, "examples", "DeepFM", "main.py"), "_diag_deepfm",
                        ["--batch-norm", "True"] + ([] if big else ["--hidden-dims", "32,24,16", "--hidden-dropout-rate", "0.0"]))
    ds = deepfm_dataset(z)
    model = mod.DeepFM(ds)
    rng = np.random.default_rng(5)
    if big:
        B = 4096
        x = np.stack([rng.integers(0, c, B) for c in z["cfg/counts"].tolist()], 1)
        y = rng.integers(0, 2, (B, 1))
    else:
        load_deepfm_golden(model, z)
        x, y = z["in/x"], z["in/labels"]
    batch = {f: torch.from_numpy(x[:, i:i + 1]) for i, f in enumerate(model.input_fields)}
    batch[model.Label], batch[model.Size] = torch.from_numpy(y), len(y)
    cfg = mod.cfg
    cfg.device, cfg.engine, cfg.monitors, cfg.which4best = "cuda:0", "module", ["LOSS"], "LOSS"
    coach = mod.CoachForDeepFM(dataset=ds, trainpipe=[batch], validpipe=None, testpipe=None, model=model, cfg=cfg)
    m = coach.get_res_sys_arch()
    spec = bridge.OptSpec(coach, m)
    ad = bridge.DeepFMAdapter(coach, m, spec)
    data = {k: (v.to(coach.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    g_ref, trace = bridge._script_step(coach, m, data)
    ad.probe_step(coach, data)
    torch.cuda.synchronize()
    grads = ad.named_grads()
    glob = max(float(g.abs().max()) for g in g_ref.values())
    print(f"global max |grad| {glob:.3e}; trace {trace}")
    # the same step in float64 on the CPU: which of the two fp32 paths is the noisy one
    import copy
    m64 = copy.deepcopy(m).cpu().double().train()
    for mm in m64.modules():
        if isinstance(mm, torch.nn.Dropout):
            mm.p = 0.0
    d64 = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batch.items()}
    loss = m64(d64)["rec_loss"]
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m64.parameters(), 10)
    g64 = {k: p.grad for k, p in m64.named_parameters()}
    for k in ad.named_views():
        ge, gr, gt = grads[k].reshape(-1).double().cpu(), g_ref[k].reshape(-1).double().cpu(), g64[k].reshape(-1)
        print(f"{k:40s} max|f64| {float(gt.abs().max()):.3e}  aten-f64 {float((gr - gt).abs().max()):.3e}  engine-f64 {float((ge - gt).abs().max()):.3e}  engine-aten {float((ge - gr).abs().max()):.3e}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "deepfm")
