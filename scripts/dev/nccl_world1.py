"""Dev check (GPU): torch.distributed's nccl (= RCCL) backend initialises on this image and runs the bench's collectives (world 1)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
u = torch.arange(4 * 1024, dtype=torch.int64, device=dev).view(4, 1024)
w = dist.all_reduce(u, async_op=True); w.wait()
t = torch.tensor([1.5], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(); torch.cuda.synchronize()
print("nccl world-1 ok:", dist.get_backend(), int(u.sum()), float(t))
dist.destroy_process_group()
