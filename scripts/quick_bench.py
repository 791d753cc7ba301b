import torch, time, numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recboard_amd import lib
if os.environ.get("RECENGINE_LIB"):   # a diagnostic build of the library
    lib.LIB_PATH = os.environ["RECENGINE_LIB"]
from recboard_amd.sasrec import SASRecEngine
torch.manual_seed(1)
N,B,S=12101,512,50
enc = sys.argv[1] if len(sys.argv) > 1 else 'fused'
m = SASRecEngine(N, 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6, encoder=enc)
g = torch.Generator().manual_seed(1)
lens = torch.clamp(torch.distributions.Geometric(probs=1/5.9).sample((B,)).long()+1, 1, 49)
seq = torch.zeros(B,S,dtype=torch.long); pos=torch.zeros_like(seq); neg=torch.zeros_like(seq)
for b in range(B):
    L=int(lens[b]); seq[b,S-L:]=torch.randint(1,N+1,(L,),generator=g); pos[b,S-L:]=torch.randint(0,N,(L,),generator=g); neg[b,S-L:]=torch.randint(0,N,(L,),generator=g)
seq,pos,neg=seq.cuda(),pos.cuda(),neg.cuda()
aux = m.batch_aux_fused(seq,pos,neg) if enc=='fused' else m.batch_aux(seq,pos,neg)
for _ in range(10): m.train_step(seq,pos,neg,aux)
torch.cuda.synchronize(); t=time.time()
for _ in range(50): l=m.train_step(seq,pos,neg,aux)
torch.cuda.synchronize(); dt=(time.time()-t)/50
print("step ms", dt*1e3, "samples/s", B/dt, "loss", l.item())
