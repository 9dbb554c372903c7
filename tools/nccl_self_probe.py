"""Does RCCL accept grouped send/recv to the own rank (single GPU)?  Used for the loopback rehearsal of the exchange."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
a = torch.arange(1000, dtype=torch.float64, device="cuda")
b = torch.zeros(1000, dtype=torch.float64, device="cuda")
ops = [dist.P2POp(dist.isend, a[:600], 0), dist.P2POp(dist.irecv, b[:600], 0)]
for r in dist.batch_isend_irecv(ops):
    r.wait()
torch.cuda.synchronize()
print("self send/recv ok:", bool((b[:600] == a[:600]).all()), float(b.sum()))
dist.destroy_process_group()
