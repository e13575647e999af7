import os, sys, datetime
import torch, torch.distributed as dist
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=60))
    t = torch.ones(4, device="cuda") * (dist.get_rank() + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", dist.get_rank(), "all_reduce ok", t.tolist(), flush=True)
except Exception as e:
    print("rank", os.environ.get("RANK"), "FAILED:", type(e).__name__, str(e)[:300], flush=True)
    sys.exit(1)
