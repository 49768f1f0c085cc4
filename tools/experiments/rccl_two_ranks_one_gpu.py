"""Does RCCL accept two ranks on ONE device?  (torchrun --nproc-per-node 2; both ranks use cuda:0)"""
import os, sys, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    x = torch.full((1024,), float(rank + 1), device="cuda:0")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    y = torch.zeros(1024, device="cuda:0")
    if rank == 0:
        dist.send(x, 1); dist.recv(y, 1)
    else:
        dist.recv(y, 0); dist.send(x, 0)
    torch.cuda.synchronize()
    print(f"[rank {rank}] all_reduce -> {x[0].item()}  recv -> {y[0].item()}  OK", flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f"[rank {rank}] FAILED: {type(e).__name__}: {str(e)[:600]}", flush=True)
    sys.exit(3)
