"""One hipGraph replay per training step for the engines whose step is a fixed chain of launches (MF-BPR, LightGCN, DeepFM): the
reference's loop body (`MF-BPR/main.py:116-123`, `LightGCN/main.py:156-164`, `DeepFM/main.py:258-268`) is ~10-40 short launches, and
issued one by one the CPU launch path, not the GPU, sets the step time.  A captured step reads its batch from static buffers and its
per-step scalars (dropout seed, Adam's step size and bias correction) from four device words written by one tiny launch (re_step_state);
everything else is the eager step's launches, recorded once."""
import contextlib
import gc

import torch

from . import ops


@contextlib.contextmanager
def recording(graph, **kw):
    """`with torch.cuda.graph(graph, **kw)` with the two guards every capture of this package needs:

    * NO CYCLIC GARBAGE COLLECTION WHILE THE STREAM RECORDS.  torch >= 2.9 no longer collects before a capture
      (`torch.compiler.config.force_cudagraph_gc` is off), and ROCm's `~CUDAGraph` ends in `hipDeviceSynchronize()`: when Python's
      collector happens to run inside the recording and frees a dead cycle that holds an earlier captured step (an engine of a finished
      epoch, a previous test's Coach), that call is refused ("operation not permitted when stream is capturing"), the error is thrown
      out of a destructor and the process dies in `std::terminate` -- SIGABRT from the main thread, "Garbage-collecting" on top of the
      traceback (gpurun_out/r5c_tests.log; the driver's round-5 GPU run).  So: collect BEFORE the recording, keep the collector off
      during it.
    * THE GRAPH OWNS WHAT IT REPLAYS: every storage whose address a launch of the recording was handed (ops._note) is kept on the
      graph object."""
    if ops._KEEP is not None:
        raise RuntimeError("recengine: nested graph captures")
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    keep = ops._KEEP = []
    try:
        with torch.cuda.graph(graph, **kw):
            yield keep
    finally:
        ops._KEEP = None
        if was_enabled:
            gc.enable()
    seen, owned = set(), []
    for st in keep:
        if st.data_ptr() not in seen:
            seen.add(st.data_ptr())
            owned.append(st)
    graph._re_owned = getattr(graph, "_re_owned", []) + owned


class CapturedStep:
    def __init__(self, body, example_inputs, restore, zeros=()):
        """body(*static_inputs, state) -> loss tensor: the step's launches, reading `state` (int32[4]: seed, 0, Adam scalars) where the
        eager step takes host scalars.  restore: tensors the warm-up run must leave as it found them (parameters, moments, running
        statistics)."""
        dev = example_inputs[0].device
        self.zeros = list(zeros)           # cleared in front of every replay (by the same launch that brings the batch in)
        self.static = [torch.empty_like(x) for x in example_inputs]
        for s, x in zip(self.static, example_inputs):
            s.copy_(x)
        self.state = torch.zeros(4, dtype=torch.int32, device=dev)
        ops.step_state(self.state, 0, 1, 1e-3)
        keep = [t.clone() for t in restore]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up: workspace allocations, lazy module loads
            body(*self.static, self.state)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with recording(self.graph, capture_error_mode="thread_local"):
            self.out = body(*self.static, self.state)
        for t, k in zip(restore, keep):
            t.copy_(k)

    def __call__(self, inputs, seed, step, lr, beta1, beta2):
        # the batch into the static buffers (int64 labels straight into an fp32 buffer), the step's scalars, the zero fills: ONE launch
        ops.stage_inputs(self.state, seed, step, lr, beta1, beta2, pairs=list(zip(self.static, inputs)), zeros=self.zeros)
        self.graph.replay()
        return self.out


def captured(engine, key, body, inputs, restore, zeros=(), static_dtypes=None):
    """The engine's captured step for this input signature (captured on first use).  static_dtypes: the dtypes of the graph's static buffers where
    they differ from the inputs' (fp32 for int64 labels: the staging launch converts on the way in); zeros: tensors that launch clears."""
    if not hasattr(engine, "_captured"):
        engine._captured = {}
    k = (key,) + tuple((tuple(x.shape), x.dtype) for x in inputs)
    if k not in engine._captured:
        ex = inputs if static_dtypes is None else tuple(x.to(d) for x, d in zip(inputs, static_dtypes))
        engine._captured[k] = CapturedStep(body, ex, restore, zeros)
    return engine._captured[k]
