"""Experiment: does a batch-preparation launch on a forked branch INSIDE the step's hipGraph run beside the step's kernels?
(If it does, the preparation of batch i+1 can ride in step i's graph and leave the critical path.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from recboard_amd import ops
from recboard_amd.sasrec import SASRecEngine

cfg = bench.BEAUTY
hb = bench.synth_batches(cfg, 8, 1)
bs = [tuple(torch.from_numpy(a).cuda() for a in b) for b in hb]


class Forked(SASRecEngine):
    fork = 0

    def _capture(self, B, S, with_adam):
        A = self.arena
        blob = torch.zeros(ops.prep_layout(B, S)[1], dtype=torch.uint8, device=self.device)
        blob2 = torch.zeros_like(blob)
        state = torch.zeros(4, dtype=torch.int32, device=self.device)
        state2 = torch.zeros_like(state)
        hyper = state.view(torch.float32)[2:4]
        z = torch.zeros((B, S), dtype=torch.int64, device=self.device)
        raw = [t.clone() for t in bs[0]]
        side2 = torch.cuda.Stream()

        def body():
            cur = torch.cuda.current_stream()
            if self.fork == 1:
                side2.wait_stream(cur)
                with torch.cuda.stream(side2):
                    ops.sasrec_batch_prep(*raw, blob=blob2, state=state2, seed=0, step=1, lr=self.lr, beta1=0.9, beta2=0.999, max_tiles=self._max_tiles())
            if self.fork == 2:   # the same launch in line (serial): what it costs when it does NOT overlap
                ops.sasrec_batch_prep(*raw, blob=blob2, state=state2, seed=0, step=1, lr=self.lr, beta1=0.9, beta2=0.999, max_tiles=self._max_tiles())
            loss = self._step_body(pb, 0, seed_dev=state)
            if with_adam:
                ops.adam_step_dev(A.data, A.grad, A.m, A.v, hyper, self.betas[0], self.betas[1], 1e-8, self.wd)
            if self.fork == 1:
                cur.wait_stream(side2)
            return loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            pb = ops.sasrec_batch_prep(z, z, z, blob=blob, state=state, seed=0, step=1, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], max_tiles=self._max_tiles())
            body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            loss = body()
        return dict(graph=graph, blob=blob, state=state, loss=loss)


for fork in (0, 1, 2, 0, 1, 2):
    m = Forked(cfg["items"], 50, 64, 2, dropout_rate=0.5, lr=5e-4, weight_decay=1e-6)
    m.fork = fork
    for i in range(20):
        m.train_step_graph(*bs[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(300):
        m.train_step_graph(*bs[i % 8])
    torch.cuda.synchronize()
    print(f"fork={fork}: {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms/step")
