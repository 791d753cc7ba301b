"""Are the captured MF-BPR / LightGCN / DeepFM steps reproducible run to run?  Two engines each, the same seeds and batches, N steps:
parameters compared bit for bit.   python scripts/determinism_models.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench_legs
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60

def mf():
    from recboard_amd.gen import MFEngine
    rng = np.random.default_rng(3)
    U, N, B = 22363, 12101, 2048
    bs = [tuple(torch.from_numpy(rng.integers(0, n, B)).cuda() for n in (U, N, N)) for _ in range(4)]
    out = []
    for rep in range(2):
        m = MFEngine(U, N, 64, lr=1e-3, weight_decay=1e-8, seed=1)
        for i in range(steps):
            m.train_step_graph(*bs[i % 4])
        out.append(m.arena.data.clone())
    return torch.equal(*out)

def lightgcn():
    from recboard_amd.gen import LightGCNEngine
    from recboard_amd.graph import to_normalized_adj
    rng = np.random.default_rng(1)
    U, N, eu, ei, wi = bench_legs.yelp_graph(rng)
    crow, col, val = to_normalized_adj(U, N, eu, ei)
    B = 2048
    bs = [tuple(torch.from_numpy(rng.integers(0, n, B)).cuda() for n in (U, N, N)) for _ in range(4)]
    out = []
    for rep in range(2):
        m = LightGCNEngine(U, N, crow, col, val, 64, 3, seed=1)
        for i in range(min(steps, 30)):
            m.train_step_graph(*bs[i % 4])
        out.append(m.arena.data.clone())
    return torch.equal(*out)

def deepfm():
    from recboard_amd.deepfm import DeepFMEngine
    counts = [94762, 25612, 7, 24, 12, 5, 50, 500, 5000, 50000]
    rng = np.random.default_rng(1)
    B = 4096
    bs = [(torch.from_numpy(np.stack([rng.integers(0, c, B) for c in counts], 1)).cuda(), torch.from_numpy((rng.random((B, 1)) < 0.3).astype(np.int64)).cuda())
          for _ in range(4)]
    out = []
    for rep in range(2):
        m = DeepFMEngine(counts, 10, (400, 400, 400), batch_norm=True, hidden_dropout_rate=0.1, lr=1e-3, embedding_decay=0.05, seed=1)
        for i in range(steps):
            m.train_step_graph(*bs[i % 4])
        out.append(m.data.clone())
    return torch.equal(*out)

for name, f in (("MF-BPR", mf), ("LightGCN", lightgcn), ("DeepFM", deepfm)):
    try:
        print(f"{name}: parameters identical after the same captured steps: {f()}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{name}: {type(e).__name__}: {e}", flush=True)
