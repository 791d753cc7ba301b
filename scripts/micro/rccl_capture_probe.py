"""Does this RCCL build capture collectives into a hipGraph?  One rank, all_reduce + all_to_all_single inside torch.cuda.graph.
Exits by itself (faulthandler) with the Python stack if anything blocks for 40 s."""
import faulthandler, os, socket, sys
faulthandler.dump_traceback_later(40, exit=True)
import torch
import torch.distributed as dist

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(1024, device="cuda")
y = torch.empty_like(x)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        dist.all_reduce(x)
        dist.all_to_all_single(y, x)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("eager collectives ok", flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "global"
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode=mode):
    dist.all_reduce(x)
    print("all_reduce captured", flush=True)
    dist.all_to_all_single(y, x)
    print("all_to_all captured", flush=True)
    z = y * 2
print("capture closed", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("replayed", float(z.sum()), flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "release":     # the graph holds the communicator's captured work: release it before the group
    del g, z
    torch.cuda.synchronize()
    print("graph released", flush=True)
dist.destroy_process_group()
print("done", flush=True)
