import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import lib
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "librecengine_dbg.so")   # the re_dbg_* switches live in the diagnostic twin (make -C recboard_amd/csrc dbg)
L = lib.load()
f = L.re_dbg_gather64
f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
f.restype = ctypes.c_int
R, n = 16 * 1024 * 1024, 4 * 1024 * 1024
W = torch.randn(R, 64, device="cuda"); idx = torch.randint(0, R, (n,), device="cuda"); out = torch.empty(n, 64, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(v, cap):
    for _ in range(3): f(W.data_ptr(), R, idx.data_ptr(), n, out.data_ptr(), v, cap, st)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10): f(W.data_ptr(), R, idx.data_ptr(), n, out.data_ptr(), v, cap, st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 10
for rnd in range(2):
    for v, name in enumerate(["ilp4", "ilp8", "ilp4nt", "ilp8nt", "ilp2", "ilp16", "ilp4nt+ntload", "ilp8nt+ntload"]):
        for cap in (2048, 4096, 16384, 65536):
            ms = run(v, cap)
            print(f"{name:14s} cap={cap:6d} {ms:.4f} ms  {n*520/ms/1e6:.0f} GB/s")
assert torch.equal(out, W[idx])
