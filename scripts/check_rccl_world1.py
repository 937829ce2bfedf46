import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
x = torch.arange(8, device="cuda", dtype=torch.int32)
parts = [torch.empty_like(x)]
dist.all_gather(parts, x)
t = torch.tensor([1.5], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(); torch.cuda.synchronize()
print("rccl world-1 ok:", parts[0].tolist(), float(t), dist.get_backend())
dist.destroy_process_group()
