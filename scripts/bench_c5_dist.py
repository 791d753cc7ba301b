"""BASELINE config 5 at N GPUs: SASRec d=128 on the synthetic 100 M-item table, the table ROW-SHARDED over the ranks
(recboard_amd.large.SASRecShardedEngine: one all-to-all round trip for the batch's rows, one all-to-all of gradient rows to their
owners + row-sparse Adam there, one all-reduce of the encoder's gradient arena).  B = 512 sequences per GPU (weak scaling).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P scripts/bench_c5_dist.py
    python scripts/bench_c5_dist.py            # one rank (a process group of size 1 over RCCL)
Rank 0 prints one JSON line; `value` is the whole job's sequences/s (max over ranks of the timed region)."""
import argparse, json, os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--items", type=int, default=100_000_000)
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--capacity-factor", type=float, default=None, help="fixed-capacity exchange (no host sync); default: exact sizes + dedup")
ap.add_argument("--graph", action="store_true", help="the whole step (both exchanges included) as one hipGraph replay; needs --capacity-factor")
ap.add_argument("--all-positions", action="store_true", help="the round-1 step: criterion over all B*S positions, one row per lookup")
args = ap.parse_args()

rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
if "MASTER_ADDR" not in os.environ:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
dist.init_process_group("nccl", rank=rank, world_size=world)

from recboard_amd.large import SASRecShardedEngine

N, D, B, S = args.items, args.dim, args.batch, 50
t0 = time.time()
model = SASRecShardedEngine(N, S, D, 2, dropout_rate=0.5, loss="BCE", lr=1e-3, weight_decay=1e-6, seed=1, device=f"cuda:{local}",
                            capacity_factor=args.capacity_factor)
if args.all_positions:
    model.compact_rows = False
torch.cuda.synchronize()
t_init = time.time() - t0
rng = np.random.default_rng(1 + rank)
batches = []
for _ in range(4):
    lens = np.clip(rng.geometric(1 / 5.9, B) + 1, 1, S - 1)
    seq = np.zeros((B, S), np.int64)
    for b in range(B):
        seq[b, S - lens[b]:] = np.minimum(rng.zipf(1.05, lens[b]), N)
    pos = np.where(seq > 0, np.minimum(rng.zipf(1.05, (B, S)), N) - 1, 0)
    neg = np.where(seq > 0, rng.integers(0, N, (B, S)), 0)
    batches.append(tuple(torch.from_numpy(a).cuda() for a in (seq, pos, neg)))
step = model.train_step_graph if args.graph else model.train_step
for i in range(args.warmup):
    step(*batches[i % 4])
dist.barrier(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(args.steps):
    loss = step(*batches[i % 4])
torch.cuda.synchronize(); dist.barrier()
if args.capacity_factor is not None:
    model.table.check_capacity()
dt = torch.tensor([time.perf_counter() - t0], device="cuda", dtype=torch.float64)
dist.all_reduce(dt, op=dist.ReduceOp.MAX)
dt = float(dt.item()) / args.steps
free, total = torch.cuda.mem_get_info()
if rank == 0:
    print(json.dumps({"metric": "train samples/sec (SASRec d=128, 100 M-item table row-sharded over the GPUs, B=512/GPU)",
                      "value": round(world * B / dt, 1), "unit": "samples/s", "n_gpus": world, "ms_per_step": round(dt * 1e3, 3),
                      "scaling": "weak", "final_loss_rank0": round(float(loss), 5), "rows_per_rank": model.table.local_rows,
                      "table_GB_per_rank": round(3 * model.table.local_rows * D * 4 / 1e9, 1), "table_init_s": round(t_init, 1),
                      "hbm_used_GB_rank0": round((total - free) / 1e9, 1),
                      "launch": ("one batch-preparation launch + one hipGraph replay" if args.graph else "eager") + (" (the exchange's split sizes are host-side)" if args.capacity_factor is None else f" (fixed-capacity exchange, factor {args.capacity_factor}: no host sync)"),
                      "step": "all positions" if args.all_positions else "compact rows (one launch for forward + criterion + backward)",
                      "world_size": dist.get_world_size(), "backend": dist.get_backend(), "encoder": f"{model.encoder} (D = {D})"}))
model.release_graphs()
dist.destroy_process_group()
