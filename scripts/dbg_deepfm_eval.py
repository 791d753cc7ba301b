"""debug: adopted DeepFM after training -- engine eval logits vs the adopted module's own forward vs a fresh module with its state_dict"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_bridge as tb
z = np.load(os.path.join(tb.G, "deepfm.npz"))
args = [a if a != "0.3" else "0.0" for a in tb.DEEPFM_ARGS]
mod = tb.import_script(os.path.join(tb.EX, "DeepFM", "main.py"), "_dbg_deepfm", args)
ds = tb.deepfm_dataset(z)
rng = np.random.default_rng(5)
counts = z["cfg/counts"].tolist()
def batches(n, B):
    return [(np.stack([rng.integers(0, c, B) for c in counts], 1), rng.integers(0, 2, (B, 1)).astype(np.float32)) for _ in range(n)]
train, valid = batches(3, 256), batches(2, 200)
def build(engine):
    model = mod.DeepFM(ds); tb.load_deepfm_golden(model, z)
    def pipe(bs):
        out = []
        for x, y in bs:
            b = {f: torch.from_numpy(x[:, i:i + 1]) for i, f in enumerate(model.input_fields)}
            b[model.Label], b[model.Size] = torch.from_numpy(y), len(y); out.append(b)
        return out
    cfg = tb._cfg(mod, engine=engine, lr=1e-2, monitors=["LOSS", "LOGLOSS", "AUC"], which4best="AUC", checkpoint_path="/tmp/dbg_ck_" + engine, epochs=2, eval_freq=1)
    return model, mod.CoachForDeepFM(dataset=ds, trainpipe=pipe(train), validpipe=pipe(valid), testpipe=None, model=model, cfg=cfg)
model, coach = build("auto")
print("engine attached:", coach._engine is not None)
for e in range(2):
    coach.train(e)
vb = coach.validpipe[0]
coach._engine.reset_ranking_buffers()
zl, _ = coach._engine.pool_logits(coach, vb)
model.eval()
with torch.no_grad():
    p_own = model({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in vb.items()}, ranking="pool").reshape(-1)
print("engine vs adopted module own forward:", float((torch.sigmoid(zl.reshape(-1)) - p_own).abs().max()))
modelm, coachm = build("module")
sd = model.state_dict()
modelm.load_state_dict(sd); modelm.eval()
with torch.no_grad():
    p_m = modelm({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in vb.items()}, ranking="pool").reshape(-1)
print("fresh module with the state_dict vs adopted module:", float((p_m - p_own).abs().max()))
for k, v in sd.items():
    w = dict(modelm.state_dict())[k]
    if not torch.equal(w.cpu(), v.cpu()):
        print("  differs after load:", k)
for i in range(3):
    b = model.dnn[i].bn
    print(i, "running_mean", float(b.running_mean.abs().mean()), "engine rm", float(coach._engine.eng.running[i][0].abs().mean()), "tracked", int(b.num_batches_tracked))
