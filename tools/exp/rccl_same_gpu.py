"""Does this RCCL accept two ranks on ONE device? (It would let the engine's own communicator run with 2 ranks on a 1-GPU box.)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/exp/rccl_same_gpu.py"""
import os
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=2)
t = torch.full((4,), float(rank + 1), device="cuda:0")
try:
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", rank, "all_reduce ->", t.tolist(), flush=True)
except Exception as e:      # noqa: BLE001
    print("rank", rank, "FAILED:", str(e)[:300], flush=True)
dist.destroy_process_group()
