"""probe: torch symmetric memory between two processes that share ONE GPU (gloo group, cuda:0 buffers)"""
import os, sys, torch, torch.distributed as dist
import torch.distributed._symmetric_memory as sm
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
try:
    t = sm.empty((4, 1024), dtype=torch.float16, device="cuda:0")
    hdl = sm.rendezvous(t, dist.group.WORLD.group_name)
    t.fill_(rank + 1)
    hdl.barrier()
    peer = hdl.get_buffer((rank + 1) % world, (4, 1024), torch.float16)
    peer[rank].copy_(t[rank] * 10)           # write my row into the peer's buffer
    hdl.barrier()
    torch.cuda.synchronize()
    print(rank, "ok", t[:, 0].tolist(), flush=True)
except Exception as e:
    print(rank, "FAILED", type(e).__name__, str(e)[:300], flush=True)
dist.destroy_process_group()
