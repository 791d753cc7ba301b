"""The two ways a captured step killed a process in round 5, each with its reproducer and the guard that closes it
(`recboard_amd.capture.recording`, the one place this package records a hipGraph):

1. Python's cyclic collector running INSIDE a recording and freeing a dead cycle that owns an earlier captured graph: ROCm's `~CUDAGraph`
   ends in `hipDeviceSynchronize()`, refused while the thread's stream captures -> exception out of a destructor -> `std::terminate` ->
   SIGABRT from the main thread ("Garbage-collecting" on top of the traceback).  torch 2.10's `torch.cuda.graph.__enter__` no longer
   collects first.  Which test dies depends on where the collector's allocation counter trips: intermittent from box to box.
2. A graph replaying the address of a tensor nobody owns any more (an SpMM plan evicted from a cache, a temporary workspace): the caching
   allocator hands the bytes to someone else, and the next capture's `empty_cache()` unmaps them -> "Memory access fault by GPU".

The scripts run in child processes: the unguarded ones are EXPECTED to die."""
import os
import signal
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PRELUDE = f"""
import gc, sys
sys.path.insert(0, {ROOT!r})
import torch
from recboard_amd.capture import recording
gc.disable()                                   # (the dead cycle below must survive until the recording: the reproducer picks the moment)
x = torch.zeros(1024, device="cuda")

def leave_a_dead_cycle_that_owns_a_captured_graph():
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        x.add_(1)
    holder = {{"graph": g}}
    holder["self"] = holder                    # an engine <-> closure cycle of a finished epoch / a previous test
leave_a_dead_cycle_that_owns_a_captured_graph()
x.zero_()
g2 = torch.cuda.CUDAGraph()
"""

UNGUARDED = _PRELUDE + """
with torch.cuda.graph(g2, capture_error_mode="thread_local"):
    x.add_(1)
    gc.collect()                               # what the automatic collector does when its counter trips inside a recording
    x.add_(1)
g2.replay(); torch.cuda.synchronize()
print("survived", float(x[0]))
"""

GUARDED = _PRELUDE + """
gc.enable()                                    # (the guard must cope with a live collector: it collects first, then switches it off)
with recording(g2, capture_error_mode="thread_local"):
    x.add_(1)
    assert not gc.isenabled()
    junk = [[i] for i in range(20000)]         # enough allocations to trip an enabled collector many times
    x.add_(1)
assert gc.isenabled()
g2.replay(); torch.cuda.synchronize()
print("survived", float(x[0]))
"""


def _run(script):
    return subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=240, cwd=ROOT)


@pytest.mark.gpu
def test_collector_inside_a_recording_kills_the_process_and_the_guard_prevents_it():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = _run(GUARDED)
    assert r.returncode == 0 and "survived 2.0" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    u = _run(UNGUARDED)
    os.write(2, f"\n[capture-guard] unguarded reproducer: rc {u.returncode}; stderr tail: {u.stderr[-400:]!r}\n".encode())
    if u.returncode == 0:
        pytest.skip("this torch / ROCm build survives a CUDAGraph destroyed inside a recording (the guard is then merely harmless)")
    assert u.returncode in (-signal.SIGABRT, 134), (u.returncode, u.stderr[-2000:])


@pytest.mark.gpu
def test_a_graph_owns_every_storage_its_launches_were_handed():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from recboard_amd import ops
    from recboard_amd.capture import recording
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    n, D = 300, 64
    dense = (torch.rand(n, n, generator=g) < 0.05).float() * torch.rand(n, n, generator=g)
    csr = dense.to_sparse_csr()
    crow, col, val = (t.to(dev).contiguous() for t in (csr.crow_indices(), csr.col_indices(), csr.values()))
    X = torch.randn(n, D, generator=g).to(dev)
    out = torch.empty_like(X)
    plan = ops.spmm_plan(crow, D)
    ops.spmm_csr(crow, col, val, plan, X, out)                       # warm-up
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with recording(graph, capture_error_mode="thread_local") as keep:
        ops.spmm_csr(crow, col, val, plan, X, out)
    assert ops._KEEP is None and keep
    owned = {st.data_ptr() for st in graph._re_owned}
    handed = [crow, col, val, X, out, plan.row_order, plan.ws] + ([plan.chunk_row, plan.chunk_ptr] if plan.nlong else [])
    assert all(t.untyped_storage().data_ptr() in owned for t in handed)
    # the plan's last reference goes away (torch_ops._PLANS.clear(), an adjacency edited in place: the round-5 fault), later captures
    # return every cached block to the driver -- the graph still replays mapped memory with the plan's bytes in it
    want = (dense.double() @ X.double().cpu()).float()
    del plan, handed
    torch.cuda.empty_cache()
    filler = [torch.full((1 << 16,), float("nan"), device=dev) for _ in range(64)]      # whatever is recycled is overwritten
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    del filler
    assert torch.allclose(out.cpu(), want, rtol=1e-5, atol=1e-5)
