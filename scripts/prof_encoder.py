"""Phase timing of the fused encoder kernels (needs a -DSE_PROFILE build of librecengine.so):
   make -C recboard_amd/csrc clean && make -C recboard_amd/csrc CXXFLAGS+=-DSE_PROFILE ... ; python scripts/prof_encoder.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from recboard_amd import lib
from recboard_amd.sasrec import SASRecEngine

cfg = bench.BEAUTY
model = SASRecEngine(cfg["items"], cfg["S"], cfg["D"], cfg["L"], dropout_rate=cfg["p_drop"], loss="BCE", seed=1)
seq, pos, neg = [torch.from_numpy(a).cuda() for a in bench.synth_batches(cfg, 1, seed=1)[0]]
aux = model.batch_aux_fused(seq, pos, neg)
print("work items:", int(aux[2][1].item()), "short of", seq.shape[0])
for _ in range(5):
    model.train_step(seq, pos, neg, aux)
torch.cuda.synchronize()
L = lib.load()
m = np.zeros((2, 64), dtype=np.int64)
for which, fn in ((0, L.re_dbg_encoder_marks_fwd), (1, L.re_dbg_encoder_marks_bwd)):
    buf = (ctypes.c_ulonglong * 64)()
    fn.restype = ctypes.c_int
    assert fn(buf) == 0
    m[which] = np.array(list(buf), dtype=np.int64)
names = {0: ["decode+load x0", "1 LN_a", "2 qkv proj", "3 scores", "softmax", "4 o=Av", "5 out_proj", "6 LN_f", "7 ffn1", "8 ffn2 (block 0 ends)", "-> lastLN (incl. block 1)", "lastLN + store"],
         1: ["decode", "load dIn(+lastLN bwd)", "pad mask/drop2", "A ffn2", "B ffn1", "C LN_f bwd", "D out_proj", "E load V,P", "E gemms dP,dV", "E softmax bwd", "F dQ,dK", "G projections", "H LN_a bwd", "store"]}
for which, label in ((0, "forward (block 0 phases)"), (1, "backward (last launched block)")):
    t = m[which]
    n = len(names[which])
    print(label, " total cycles", t[n] - t[0] if which == 1 else t[n - 1] - t[0])
    for i in range(n if which == 1 else n - 1):
        print(f"   {names[which][i]:32s} {int(t[i + 1] - t[i]):8d}")
