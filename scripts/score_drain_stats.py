"""Drain statistics of the split score kernel at Beauty's shape (diagnostic twin: make -C recboard_amd/csrc dbg): drains, insertion rounds,
queue entries per call; per wave-tile averages.    python scripts/score_drain_stats.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recboard_amd import lib, ops  # noqa: E402
lib.LIB_PATH = os.path.join(os.path.dirname(lib.LIB_PATH), "librecengine_dbg.so")
L = lib.load()
L.re_dbg_score_diag.argtypes = [ctypes.c_int]; L.re_dbg_score_diag.restype = None
L.re_dbg_score_counters.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.re_dbg_score_counters.restype = None
U, N = 22363, 12101
g = torch.Generator(device="cuda").manual_seed(11)
q = torch.randn(U, 64, device="cuda", generator=g)
E = torch.randn(N, 64, device="cuda", generator=g)
sp = torch.arange(0, U + 1, device="cuda") * 8
si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
prep = ops.score_prepare(E)
ops.score_topk(q, E, sp, si, 50, prep=prep)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 4)()
L.re_dbg_score_counters(out, 1)
L.re_dbg_score_diag(3)
ops.score_topk(q, E, sp, si, 50, prep=prep)
torch.cuda.synchronize()
L.re_dbg_score_counters(out, 1)
L.re_dbg_score_diag(0)
drains, rounds, entries = int(out[0]), int(out[1]), int(out[2])
tiles = ((U + 31) // 32) * ((N + 31) // 32)
print(f"drains {drains}  rounds {rounds}  queue entries {entries}  | wave-tiles {tiles}: {drains / tiles:.3f} drains, {rounds / tiles:.3f} rounds, "
      f"{entries / tiles:.2f} entries per wave-tile; entries per user {entries / U:.1f}; rounds per drain {rounds / max(drains, 1):.2f}")
