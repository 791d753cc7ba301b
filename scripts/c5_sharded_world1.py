import os, sys, json
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import bench_legs
print(json.dumps(bench_legs.config5_sharded(dist, 0, 1, 0, steps=20, warmup=4, nbatch=8)))
dist.destroy_process_group()
