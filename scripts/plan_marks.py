"""Where the plan workgroup of the batch-preparation launch spends its time (make encprof: shader-clock stamps, 10 ns units)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "librecengine_encprof.so")
import bench
from recboard_amd import ops
from recboard_amd.sasrec import SASRecEngine
cfg = bench.BEAUTY
m = SASRecEngine(cfg["items"], 50, 64, 2, dropout_rate=0.5)
bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in bench.synth_batches(cfg, 4, 1)]
blob = torch.zeros(ops.prep_layout(512, 50)[1], dtype=torch.uint8, device="cuda")
state = torch.zeros(4, dtype=torch.int32, device="cuda")
L = lib.load()
L.re_dbg_plan_marks.argtypes = [ctypes.c_void_p]
for rep in range(3):
    for b in bs:
        ops.sasrec_batch_prep(*b, blob=blob, state=state, weights=m._prep_weights(512, 50))
        torch.cuda.synchronize()
        out = (ctypes.c_uint64 * 4)()
        L.re_dbg_plan_marks(out)
        print("spans %.2f us  ranks %.2f us  row map %.2f us" % (out[1] / 100, (out[2] - out[1]) / 100, (out[3] - out[2]) / 100))
